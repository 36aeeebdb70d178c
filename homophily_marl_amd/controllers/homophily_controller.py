"""HomophilyMAC: multi-agent controller with an env head and an incentive head per agent.

Surface of the reference src/controllers/homophily_controller.py:11-206 (select_actions_env / select_actions_inc /
forward / init_hidden / parameters* / load_state / save_models / load_models / cuda).  The agent-input assembly of
`_build_inputs` (:127-184) writes straight into one preallocated [B*n, input_shape] buffer: the conv encoder output
goes into its first columns and the HIP kernel ssd_build_inputs fills the rest (one launch instead of ~15 torch ops).
"""
import numpy as np
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from ..components.action_selectors import REGISTRY as action_REGISTRY
from ..modules.agents import REGISTRY as agent_REGISTRY
from .. import ops


class HomophilyMAC(nn.Module):
    def __init__(self, scheme, groups, args):
        super().__init__()
        self.n_agents = args.n_agents
        self.args = args
        # the shipped flag set (config/default.yaml:45-51) has a HIP kernel for the input tail and the fused rollout kernels;
        # any other combination of the _build_inputs flags is assembled with torch expressions
        self.shipped_flags = bool(args.obs_last_action and args.obs_agent_id and args.obs_reward and args.obs_inc_reward
                                  and args.obs_agent_pos and not getattr(args, "obs_others_last_action", False)
                                  and not getattr(args, "obs_distance", False))
        # the same flags as the ssd_policy_head.input_flags word of the fused rollout heads (None: obs_others_last_action, whose
        # n * n_actions columns do not fit the heads' 64-column weight image -> the generic captured timestep)
        self.input_flags = None if getattr(args, "obs_others_last_action", False) else (
            1 * bool(args.obs_last_action) | 2 * bool(args.obs_agent_id) | 4 * bool(args.obs_reward) | 8 * bool(args.obs_inc_reward)
            | 16 * bool(getattr(args, "obs_distance", False)) | 32 * bool(args.obs_agent_pos))
        # all seven blocks as ssd_build_inputs_flags bits (the learner-side input assembly on the device takes any set)
        self.input_flags_all = ((self.input_flags if self.input_flags is not None else
                                 (1 * bool(args.obs_last_action) | 2 * bool(args.obs_agent_id) | 4 * bool(args.obs_reward)
                                  | 8 * bool(args.obs_inc_reward) | 16 * bool(getattr(args, "obs_distance", False)) | 32 * bool(args.obs_agent_pos)))
                                | 64 * bool(getattr(args, "obs_others_last_action", False)))
        self.input_shape = self._get_input_shape(scheme)
        self.agent = agent_REGISTRY[args.agent](self.input_shape, args)
        self.agent_output_type = args.agent_output_type
        self.action_selector = action_REGISTRY[args.action_selector](args)
        self.h_env = self.h_inc = None
        self.extra_return_env = self.extra_return_inc = None
        self.pos_scale = float(np.linalg.norm(args.state_dims))
        self.register_buffer("inc_mask_actions", (1 - th.eye(self.n_agents)).reshape(1, self.n_agents, self.n_agents, 1), persistent=False)

    # ---- acting ---------------------------------------------------------------------------------------------
    def select_actions_env(self, ep_batch, t_ep, t_env, bs=slice(None), test_mode=False):
        avail = ep_batch["avail_actions"][:, t_ep]
        q_env = self.forward_env(ep_batch, t_ep, test_mode=test_mode)
        chosen = self.action_selector.select_action(q_env[bs], avail[bs], t_env, test_mode=test_mode)
        return chosen.unsqueeze(-1)                                                   # [bs, n, 1]

    def select_actions_inc(self, chosen_actions, ep_batch, t_ep, t_env, bs=slice(None), test_mode=False, agent_pos_replay=None):
        q_inc = self.forward_inc(ep_batch, t_ep, chosen_actions, test_mode=test_mode)
        chosen = self.action_selector.select_action(q_inc[bs], th.ones_like(q_inc[bs]), t_env, test_mode=test_mode)
        return chosen.unsqueeze(-1) * self.inc_mask_actions.to(chosen.dtype)          # no self-incentive; [bs, n, n, 1]

    # ---- network evaluation ---------------------------------------------------------------------------------
    def forward_env(self, ep_batch, t, test_mode=False, learning_mode=False):
        self.agent_inputs = self._build_inputs(ep_batch, t)
        q_env, self.h_env, self.extra_return_env = self.agent.forward_env(self.agent_inputs, self.h_env, learning_mode)
        return q_env.reshape(ep_batch.batch_size, self.n_agents, -1)

    def forward_inc(self, ep_batch, t, chosen_actions, test_mode=False, learning_mode=False):
        actions_env = F.one_hot(chosen_actions.squeeze(-1), num_classes=self.args.n_actions)
        q_inc, self.h_inc, self.extra_return_inc = self.agent.forward_inc(
            self.agent_inputs, self.h_inc, actions_env,
            ep_batch["agent_pos"][:, t] / self.pos_scale, ep_batch["agent_orientation"][:, t],
            ep_batch["reward"][:, t].unsqueeze(-1), ep_batch["clean_num"][:, t].unsqueeze(-1),
            ep_batch["apple_den"][:, t].unsqueeze(-1), learning_mode=learning_mode)
        return q_inc.reshape(ep_batch.batch_size, self.n_agents, self.n_agents, -1)

    def forward(self, ep_batch, t, test_mode=False):
        """Learning-time evaluation: env head on the stored inputs, inc head on the stored env actions."""
        q_env = self.forward_env(ep_batch, t, test_mode=test_mode, learning_mode=True)
        q_inc = self.forward_inc(ep_batch, t, ep_batch["actions"][:, t, :], test_mode=test_mode, learning_mode=True)
        return q_env, q_inc, self.extra_return_inc

    def init_hidden(self, batch_size):
        h_env, h_inc = self.agent.init_hidden()
        self.h_env = h_env.repeat(batch_size, 1, 1, 1)
        self.h_inc = h_inc.repeat(batch_size, 1, 1, 1)

    # ---- parameters / persistence ---------------------------------------------------------------------------
    def parameters(self, recurse=True):
        return self.agent.parameters()

    def parameters_env(self):
        return self.agent.parameters_env()

    def parameters_inc(self):
        return self.agent.parameters_inc()

    def load_state(self, other_mac):
        self.agent.load_state_dict(other_mac.agent.state_dict())

    def cuda(self, device=None):
        self.agent.cuda(device)
        self.inc_mask_actions = self.inc_mask_actions.cuda(device)
        return self

    def save_models(self, path):
        th.save(self.agent.state_dict(), "{}/agent.th".format(path))

    def load_models(self, path):
        self.agent.load_state_dict(th.load("{}/agent.th".format(path), map_location=lambda storage, loc: storage, weights_only=True))

    # ---- inputs ---------------------------------------------------------------------------------------------
    @staticmethod
    def expand_codes(codes):
        """u8 class codes [..., V, V] -> f32 planes [..., 3, V, V] (ops.expand_codes)."""
        return ops.expand_codes(codes)

    def encode_obs(self, obs):
        """obs [B, n, 3, V, V] (any float dtype; or u8 class codes [B, n, V, V]) -> conv features [B * n, obs_dim_net]."""
        if obs.dtype == th.uint8:
            obs = self.expand_codes(obs)
        B = obs.shape[0]
        return self.agent.rgb_preprocess(obs.reshape(B * self.n_agents, 3, self.args.obs_dims[0], self.args.obs_dims[1]).float())

    def assemble_inputs(self, feat, last_actions, last_reward, last_actions_inc, pos, t0):
        """[feat | onehot(last action) | onehot(id) | sign(r) | sign(received incentives) | others' last actions | 1 - distances |
        pos / ||(H,W)||], each block present iff its flag is set (homophily_controller.py:137-184).  last_actions = -1 means
        "no previous step" (all-zero one-hot); t0: the whole batch is at t = 0."""
        a = self.args
        n, A = self.n_agents, a.n_actions
        B = pos.shape[0]
        F0 = feat.shape[1]
        if self.shipped_flags or feat.is_cuda:      # the device kernel builds any flag set; on the CPU the shipped one has a fast path
            if self.input_shape == F0:
                return feat
            tail = th.empty(B * n, self.input_shape - F0, dtype=th.float32, device=feat.device)
            ops.build_inputs_tail(tail, 0, last_actions, last_reward, last_actions_inc, pos, self.pos_scale, A, t0,
                                  flags=None if self.shipped_flags else self.input_flags_all)
            return th.cat([feat, tail], dim=1)
        dev = feat.device
        if t0:
            last_actions = th.full((B, n), -1, dtype=th.long, device=dev)
            last_reward = th.zeros(B, n, device=dev)
            last_actions_inc = th.zeros(B, n, n, dtype=th.long, device=dev)
        had = (last_actions >= 0).unsqueeze(-1).float()                               # a previous step exists
        onehot = F.one_hot(last_actions.clamp(min=0), A).float() * had                # [B, n, A]
        parts = [feat.reshape(B, n, F0)]
        if a.obs_last_action:
            parts.append(onehot)
        if a.obs_agent_id:
            parts.append(th.eye(n, device=dev).unsqueeze(0).expand(B, -1, -1))
        if a.obs_reward:
            parts.append(th.sign(last_reward.float()).unsqueeze(-1))
        if a.obs_inc_reward:
            m = last_actions_inc * self.inc_mask_actions.squeeze(-1).to(last_actions_inc.dtype)   # [B, giver, receiver]
            recv = (m == 1).sum(dim=1) - (m == 2).sum(dim=1)                          # [B, receiver]
            parts.append(th.sign(recv.float()).unsqueeze(-1))
        if getattr(a, "obs_others_last_action", False):                               # every agent sees all last actions, agent order
            parts.append(onehot.reshape(B, 1, n * A).expand(-1, n, -1))
        if getattr(a, "obs_distance", False):
            parts.append(1.0 - (pos.unsqueeze(2) - pos.unsqueeze(1)).norm(dim=-1) / self.pos_scale)
        if a.obs_agent_pos:
            parts.append(pos / self.pos_scale)
        return th.cat([x.reshape(B * n, -1).float() for x in parts], dim=1)

    def unroll(self, batch):
        """q_env [B, T, n, A], q_inc [B, T, n, n, 3] for every timestep of `batch` -- what calling forward(batch, t) for
        t = 0..T-1 from fresh hidden states returns (the learner's loops, homophily_learner.py:68-91) -- with the encoder,
        the input assembly and all non-recurrent layers evaluated once over all T."""
        shared = self.unroll_shared(batch)
        parts, wh, bh = self.unroll_pre(batch, shared)
        he, hi = ops.gru_sequence_parts(parts, batch.max_seq_length, batch.batch_size, wh, bh)
        return self.agent.unroll_post(he, hi, shared["other"])

    def unroll_shared(self, batch):
        """Everything of an unroll that does not depend on the weights (the learner evaluates the live and the target net on the
        same batch): the observation as float planes, the history-shifted controller inputs, one-hot actions, the incentive
        head's per-receiver features."""
        a = self.args
        B, T, n = batch.batch_size, batch.max_seq_length, self.n_agents
        obs = batch["obs"]
        codes = None
        if obs.dtype == th.uint8:                                                      # compact storage (class codes)
            if a.rgb_input and self.agent.encoder_kernel_shape and ops.encode_codes_supported(obs):
                codes = obs                        # the encoder kernel reads the codes themselves (ops.encode_codes)
            else:
                ops._leaving_kernels("encode_codes", obs, "no encoder kernel for %s windows" % (tuple(obs.shape[-2:]),))
                obs = self.expand_codes(obs)
        acts = batch["actions"].squeeze(-1)                                            # [B, T, n]
        # history features of step t come from t - 1; at t = 0 they are zero: action -1 has an all-zero one-hot
        prev = lambda x, fill: th.cat([th.full_like(x[:, :1], fill), x[:, :-1]], dim=1)
        on_dev = batch["reward"].is_cuda
        kernel_tail = (self.shipped_flags or on_dev) and a.rgb_input and self.input_shape > a.obs_dim_net
        # the device kernel shifts the history itself (seq_len = T); the tensor-op assembly takes shifted copies
        hist = None if (kernel_tail and on_dev) else (
            prev(acts, -1).reshape(B * T, n), prev(batch["reward"], 0).reshape(B * T, n),
            prev(batch["actions_inc"].squeeze(-1), 0).reshape(B * T, n, n), batch["agent_pos"].reshape(B * T, n, 2))
        sh = dict(obs=None if codes is not None else (obs.float() if a.rgb_input else obs), codes=codes, hist=hist, tail=None)
        if on_dev and acts.dtype == th.long:
            # one launch: the incentive head's per-receiver features and the one-hot actions in the agent-major layout fc1_inc reads
            sh["onehot"] = None
            sh["other"], sh["act_tm"] = ops.unroll_other(acts, batch["agent_pos"], batch["agent_orientation"], batch["reward"], batch["clean_num"],
                                                         batch["apple_den"], self.pos_scale, a.n_actions)
        else:
            onehot = F.one_hot(acts, num_classes=a.n_actions)
            sh.update(onehot=onehot, act_tm=None,
                      other=self.agent.unroll_other(onehot, batch["agent_pos"] / self.pos_scale, batch["agent_orientation"], batch["reward"],
                                                    batch["clean_num"], batch["apple_den"], th.float32))
        if kernel_tail:
            # the non-visual input columns do not depend on the weights either
            tail = th.empty(B * T * n, self.input_shape - a.obs_dim_net, dtype=th.float32, device=batch["obs"].device)
            flags = None if self.shipped_flags else self.input_flags_all
            if hist is None:
                ops.build_inputs_tail(tail, 0, acts.reshape(B * T, n), batch["reward"].reshape(B * T, n),
                                      batch["actions_inc"].squeeze(-1).reshape(B * T, n, n), batch["agent_pos"].reshape(B * T, n, 2),
                                      self.pos_scale, a.n_actions, False, flags=flags, seq_len=T)
            else:
                ops.build_inputs_tail(tail, 0, hist[0], hist[1], hist[2], hist[3], self.pos_scale, a.n_actions, False, flags=flags)
            sh["tail"] = tail
        return sh

    def unroll_pre(self, batch, shared):
        """Encoder, input assembly, fc1 and the input-side GRU projections of this net: [gi_env, gi_inc] (each [n, T * B, 3H]), wh, bh."""
        a = self.args
        B, T, n = batch.batch_size, batch.max_seq_length, self.n_agents
        obs = shared["obs"]
        if shared.get("codes") is not None:
            feat = self.agent.rgb_preprocess_codes(shared["codes"].reshape(B * T * n, a.obs_dims[0], a.obs_dims[1]))
        elif a.rgb_input:
            feat = self.agent.rgb_preprocess(obs.reshape(B * T * n, 3, a.obs_dims[0], a.obs_dims[1]))
        else:
            feat = obs.reshape(B * T * n, -1)
        if shared["tail"] is not None:
            inputs = th.cat([feat, shared["tail"]], dim=1).reshape(B, T, n, -1)
        else:
            inputs = self.assemble_inputs(feat, *shared["hist"], False).reshape(B, T, n, -1)
        return self.agent.unroll_pre(inputs, shared["onehot"], act_tm=shared.get("act_tm"))

    def _build_inputs(self, batch, t):
        if self.args.rgb_input:
            feat = self.encode_obs(batch["obs"][:, t])
        else:
            feat = batch["obs"][:, t].reshape(batch.batch_size * self.n_agents, -1)
        if t == 0:
            return self.assemble_inputs(feat, None, None, None, batch["agent_pos"][:, t], True)
        return self.assemble_inputs(feat, batch["actions"][:, t - 1].squeeze(-1), batch["reward"][:, t - 1],
                                    batch["actions_inc"][:, t - 1].squeeze(-1), batch["agent_pos"][:, t], False)

    def _get_input_shape(self, scheme):
        a = self.args
        shape = a.obs_dim_net if a.rgb_input else scheme["obs"]["vshape"]
        if a.obs_last_action:
            shape += scheme["actions_onehot"]["vshape"][0]
        if a.obs_agent_id:
            shape += self.n_agents
        shape += int(bool(a.obs_reward)) + int(bool(a.obs_inc_reward))
        if getattr(a, "obs_others_last_action", False):
            shape += scheme["actions_onehot"]["vshape"][0] * self.n_agents
        if getattr(a, "obs_distance", False):
            shape += self.n_agents
        if a.obs_agent_pos:
            shape += 2
        return shape
