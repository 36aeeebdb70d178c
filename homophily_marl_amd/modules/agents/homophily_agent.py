"""Q-network of the homophily algorithm: a shared conv encoder and, PER AGENT (unshared weights), an env head and an
incentive head, each fc1 -> GRU cell -> dueling Q.  Behaviour and parameter names follow the reference
src/modules/agents/homophily_agent.py:7-214 so checkpoints (`agent.th`) interchange; the evaluation is re-organised
for the GPU: the per-agent weights [1, n, in, out] are applied as ONE batched GEMM over the agent dimension
([n, B, in] x [n, in, out], hipBLASLt / MFMA) and the six GRU projections run as two GEMMs on concatenated weights.
"""
import math

import torch as th
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


class HomophilyAgent(nn.Module):
    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.n_agents = n = args.n_agents
        self.n_actions = args.n_actions
        self.n_inc_actions = args.n_inc_actions
        self.input_shape = input_shape
        self.hidden = H = args.rnn_hidden_dim
        self.extra_input_shape = args.n_actions + 2 + 2 + 3   # action one-hot, pos, orientation, reward/clean_num/apple_den
        self._gru_cache = None
        self._fc2_cache = None
        if args.rgb_input:
            k = args.conv_kernel
            flat = args.conv_out * (args.obs_dims[0] - k + 1) * (args.obs_dims[1] - k + 1)
            self.conv_to_fc = nn.Sequential(nn.Conv2d(3, args.conv_out, k, args.conv_stride), nn.LeakyReLU(), nn.Flatten(),
                                            nn.Linear(flat, args.obs_dim_net), nn.LeakyReLU())
        # the encoder kernel (ops.encode_codes) is instantiated for the shipped encoder: Conv2d(3, 6, 3, 1) + Linear(., 32)
        self.encoder_kernel_shape = bool(args.rgb_input) and (args.conv_out, args.conv_kernel, args.conv_stride, args.obs_dim_net) == (6, 3, 1, 32)

        def weight(fan_in, fan_out, kaiming=False):
            t = th.empty(1, n, fan_in, fan_out)
            if kaiming:     # reference: kaiming_uniform_(a=sqrt(5)) on the 4-D tensor (homophily_agent.py:29-30)
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            else:
                nn.init.uniform_(t, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            return nn.Parameter(t)

        def bias(fan_in, fan_out):
            return nn.Parameter(nn.init.uniform_(th.empty(1, n, 1, fan_out), -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in)))

        # registration order = reference order (homophily_agent.py:37-125); optimiser state interchange depends on it
        for head, fc1_in, fc2_in, fc2_out in (("env", input_shape, H, self.n_actions),
                                              ("inc", input_shape + self.n_actions, H + self.extra_input_shape, self.n_inc_actions)):
            setattr(self, "fc1_%s_w" % head, weight(fc1_in, H, kaiming=True))
            setattr(self, "fc1_%s_b" % head, bias(fc1_in, H))
            for gate in ("ir", "hr", "iz", "hz", "in", "hn"):
                setattr(self, "rnn_%s_%s_w" % (head, gate), weight(H, H))
                setattr(self, "rnn_%s_%s_b" % (head, gate), bias(H, H))
            setattr(self, "fc2_%s_w" % head, weight(fc2_in, fc2_out, kaiming=True))
            setattr(self, "fc2_%s_b" % head, bias(fc2_in, fc2_out))
            setattr(self, "fc2_%s_v_w" % head, weight(fc2_in, 1, kaiming=True))
            setattr(self, "fc2_%s_v_b" % head, bias(fc2_in, 1))

    # ---- parameter groups (homophily_agent.py:127-146): the encoder belongs to BOTH groups -----------------------
    def _group(self, tag):
        params = list(self.conv_to_fc.parameters()) if self.args.rgb_input else []
        return params + [p for name, p in self.named_parameters() if tag in name]

    def parameters_env(self):
        return self._group("env")

    def parameters_inc(self):
        return self._group("inc")

    def init_hidden(self):
        z = self.fc1_env_w.new_zeros(1, self.n_agents, 1, self.hidden)
        return z.detach(), z.clone().detach()

    def rgb_preprocess(self, x):
        """Conv2d + LeakyReLU + Flatten + Linear + LeakyReLU (homophily_agent.py:20-27,213-214).  With gradients on the device the
        two bias gradients are
        evaluated as GEMMs (ops.column_sums: ATen's multi-block reductions are not safe inside a replayed hipGraph here)."""
        if x.is_cuda and th.is_grad_enabled():
            c = self.conv_to_fc
            y = ops.channel_bias(F.conv2d(x, c[0].weight, None, c[0].stride), c[0].bias)
            return c[4](ops.bias_linear(c[2](c[1](y)), c[3].weight, c[3].bias))
        return self.conv_to_fc(x)

    def rgb_preprocess_codes(self, codes):
        """rgb_preprocess of windows stored as u8 class codes [R, V, V] on the device (obs_storage: code), through the encoder
        kernel (ops.encode_codes) instead of expanding them to f32 planes first."""
        c = self.conv_to_fc
        return ops.encode_codes(codes, c[0].weight, c[0].bias, c[3].weight, c[3].bias)

    # ---- building blocks ----------------------------------------------------------------------------------------
    # squeeze, not [0]: the backward of a select materialises a zero tensor + copy per parameter (36 of them per step)
    def _w(self, name):
        return getattr(self, name).squeeze(0)    # [n, in, out]

    def _b(self, name):
        return getattr(self, name).squeeze(0)    # [n, 1, out]

    def _gru(self, head, x, h):
        """x, h: [n, B, H].  r, z, n gates exactly as homophily_agent.py:162-165 / 188-191."""
        p = "rnn_%s_" % head
        wi = th.cat([self._w(p + "ir_w"), self._w(p + "iz_w"), self._w(p + "in_w")], dim=2)
        wh = th.cat([self._w(p + "hr_w"), self._w(p + "hz_w"), self._w(p + "hn_w")], dim=2)
        bi = th.cat([self._b(p + "ir_b"), self._b(p + "iz_b"), self._b(p + "in_b")], dim=2)
        bh = th.cat([self._b(p + "hr_b"), self._b(p + "hz_b"), self._b(p + "hn_b")], dim=2)
        gi = th.baddbmm(bi, x, wi)
        gh = th.baddbmm(bh, h, wh)
        H = self.hidden
        r = th.sigmoid(gi[..., :H] + gh[..., :H])
        z = th.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
        cand = th.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
        return (1 - z) * cand + z * h

    # ---- whole-sequence evaluation (learner) -------------------------------------------------------------------
    def _gru_weights(self, head):
        """(W_i, W_h, b_i, b_h) of a head with the r, z, n blocks side by side.  A frozen copy of the net (the learner's target
        network: no parameter requires grad) keeps them in buffers that are refreshed in place whenever its weights are loaded
        (load_state_dict) -- 8 concatenations per evaluation otherwise, and captured graphs keep reading the same addresses."""
        return self._gru_weights_both()[0 if head == "env" else 1]

    def _gru_weights_both(self):
        """_gru_weights of the env and the inc head; a live net packs all eight images with ONE launch (ops.cat_groups: 8
        concatenations forward, 24 strided gradient copies backward otherwise)."""
        frozen = not self.rnn_env_ir_w.requires_grad
        if frozen:
            if self._gru_cache is None:
                got = {h: self._gru_weights_cat(h) for h in ("env", "inc")}
                if th.is_grad_enabled():
                    return got["env"], got["inc"]
                self._gru_cache = got
            return self._gru_cache["env"], self._gru_cache["inc"]
        groups = []
        for head in ("env", "inc"):
            p = "rnn_%s_" % head
            groups += [[self._w(p + g + "_w") for g in ("ir", "iz", "in")], [self._w(p + g + "_w") for g in ("hr", "hz", "hn")],
                       [self._b(p + g + "_b") for g in ("ir", "iz", "in")], [self._b(p + g + "_b") for g in ("hr", "hz", "hn")]]
        out = ops.cat_groups(groups)
        return tuple(out[:4]), tuple(out[4:])

    def _gru_weights_cat(self, head):
        p = "rnn_%s_" % head
        wi = th.cat([self._w(p + "ir_w"), self._w(p + "iz_w"), self._w(p + "in_w")], dim=2)
        wh = th.cat([self._w(p + "hr_w"), self._w(p + "hz_w"), self._w(p + "hn_w")], dim=2)
        bi = th.cat([self._b(p + "ir_b"), self._b(p + "iz_b"), self._b(p + "in_b")], dim=2)
        bh = th.cat([self._b(p + "hr_b"), self._b(p + "hz_b"), self._b(p + "hn_b")], dim=2)
        return wi, wh, bi, bh

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        if self._gru_cache is not None:          # refresh in place: a captured train step reads these buffers
            with th.no_grad():
                for h, bufs in self._gru_cache.items():
                    for dst, src in zip(bufs, self._gru_weights_cat(h)):
                        dst.copy_(src)
        if self._fc2_cache is not None:
            with th.no_grad():
                bufs, self._fc2_cache = self._fc2_cache, None
                for dst, src in zip(bufs, self._fc2_merged()):
                    dst.copy_(src)
                self._fc2_cache = bufs
        return out

    def _apply(self, fn, *args, **kwargs):
        self._gru_cache = None                   # .cuda() / .to(): the caches are rebuilt on the new device at the next use
        self._fc2_cache = None
        return super()._apply(fn, *args, **kwargs)

    def unroll(self, inputs, act_onehot, agent_pos, agent_orientation, reward, clean_num, apple_den):
        """Evaluate both heads on whole episodes: inputs [B, T, n, in], act_onehot [B, T, n, A] (the env action taken at
        t), agent_pos / agent_orientation [B, T, n, 2], reward / clean_num / apple_den [B, T, n].
        Returns q_env [B, T, n, A], q_inc [B, T, n, n, 3] -- the values forward_env / forward_inc produce step by step
        (homophily_agent.py:154-208) from zero hidden states.  Everything that does not depend on the recurrence (fc1,
        the input-side GRU projections, both dueling heads) runs ONCE over all T; the recurrence itself (h @ W_h and the gate
        arithmetic, env and inc head together as 2n weight sets) is one sequence kernel per direction on the GPU.
        = unroll_pre -> ops.gru_sequence -> unroll_post (the learner runs the live and the target net through ONE sequence launch)."""
        parts, wh, bh = self.unroll_pre(inputs, act_onehot)
        he, hi = ops.gru_sequence_parts(parts, inputs.shape[1], inputs.shape[0], wh, bh)   # 2 x [n, T, B, H]: one launch for the T steps
        return self.unroll_post(he, hi, self.unroll_other(act_onehot, agent_pos, agent_orientation, reward, clean_num, apple_den, inputs.dtype))

    def unroll_pre(self, inputs, act_onehot, act_tm=None):
        """fc1 + the input-side GRU projections of both heads over all T: [gi_env, gi_inc], each [n, T * B, 3H] (set-major, rows
        t * B + b: ops.gru_sequence_parts takes them as they are), and the recurrence weights as parts [wh_env, wh_inc] (each [n, H, 3H]),
        [bh_env, bh_inc] (each [n, 1, 3H])."""
        B, T, n = inputs.shape[0], inputs.shape[1], self.n_agents
        H = self.hidden
        tm = lambda x: x.permute(2, 1, 0, 3).reshape(n, T * B, x.shape[-1])          # time-major rows [n, T*B, f]
        x = tm(inputs)
        act = act_tm if act_tm is not None else tm(act_onehot.to(inputs.dtype))   # act_tm: the one-hot actions already agent-major (ops.unroll_other)
        xe = ops.bias_bmm(x, self._w("fc1_env_w"), self._b("fc1_env_b"), leaky=True)             # fc1 + LeakyReLU as one launch
        xi = ops.bias_bmm(th.cat([x, act], dim=-1), self._w("fc1_inc_w"), self._b("fc1_inc_b"), leaky=True)
        (wie, whe, bie, bhe), (wii, whi, bii, bhi) = self._gru_weights_both()
        return [ops.bias_bmm(xe, wie, bie), ops.bias_bmm(xi, wii, bii)], [whe, whi], [bhe, bhi]      # (weight parts: no concatenation)

    @staticmethod
    def unroll_other(act_onehot, agent_pos, agent_orientation, reward, clean_num, apple_den, dtype):
        """other_j = [onehot(a_j), pos_j, orient_j, r_j, clean_j, apple_den_j] (homophily_agent.py:194-201) as [T*B, n(j), E];
        independent of the weights (the learner shares it between the live and the target net)."""
        T, B, n = act_onehot.shape[1], act_onehot.shape[0], act_onehot.shape[2]
        return th.cat([act_onehot.to(dtype), agent_pos, agent_orientation, reward.unsqueeze(-1), clean_num.unsqueeze(-1),
                       apple_den.unsqueeze(-1)], dim=-1).permute(1, 0, 2, 3).reshape(T * B, n, -1)

    def _fc2_merged(self):
        """(w_env [n, H, A + 1], b_env [n, 1, A + 1], w_inc [n, H + E, 4], b_inc [n, 1, 4]): the advantage and the value layer of each
        dueling head side by side (one layer, one launch per head and direction); ONE launch for the four concatenations (a frozen
        copy of the net -- the learner's target network -- keeps them in buffers that load_state_dict refreshes)."""
        frozen = not self.fc2_env_w.requires_grad
        if frozen and self._fc2_cache is not None:
            return self._fc2_cache
        groups = [[self._w("fc2_env_w"), self._w("fc2_env_v_w")], [self._b("fc2_env_b"), self._b("fc2_env_v_b")],
                  [self._w("fc2_inc_w"), self._w("fc2_inc_v_w")], [self._b("fc2_inc_b"), self._b("fc2_inc_v_b")]]
        out = tuple(ops.cat_groups(groups))
        if frozen and not th.is_grad_enabled():
            self._fc2_cache = out
        return out

    def unroll_post(self, he, hi, other):
        """Both dueling heads on the recurrence states of the env and the inc head (each [n, T, B, H]) and other [T*B, n(j), E]:
        q_env [B, T, n, A], q_inc [B, T, n, n, 3].  Per head ONE layer ([advantage | value] columns) and one dueling launch; the
        incentive head's rows [h_i | other_j] (homophily_agent.py:194-201) are read from the two tensors, never materialised."""
        n, H = self.n_agents, self.hidden
        T, B = he.shape[1], he.shape[2]
        he, hi = he.reshape(n, T * B, H), hi.reshape(n, T * B, H)
        w_env, b_env, w_inc, b_inc = self._fc2_merged()
        q_env = ops.dueling_head(he, None, w_env, b_env, B, T, 1)                      # v + a - mean(a) as [B, T, n, A]
        q_inc = ops.dueling_head(hi, other, w_inc, b_inc, B, T, n)                     # [B, T, n(i), n(j), 3]
        return q_env, q_inc

    # ---- heads --------------------------------------------------------------------------------------------------
    def forward_env(self, inputs, h_in, learning_mode=False):
        n, H = self.n_agents, self.hidden
        x = inputs.reshape(-1, n, self.input_shape).transpose(0, 1)                 # [n, B, in]
        h = h_in.reshape(-1, n, H).transpose(0, 1)
        x = F.leaky_relu(th.baddbmm(self._b("fc1_env_b"), x, self._w("fc1_env_w")))
        h_out = self._gru("env", x, h)
        a = th.baddbmm(self._b("fc2_env_b"), h_out, self._w("fc2_env_w"))          # [n, B, A]
        v = th.baddbmm(self._b("fc2_env_v_b"), h_out, self._w("fc2_env_v_w"))      # [n, B, 1]
        q = v + a - a.mean(dim=-1, keepdim=True)                                    # dueling (homophily_agent.py:168-170)
        return q.transpose(0, 1), h_out.transpose(0, 1).unsqueeze(2), {}
        # [B, n, A], [B, n, 1, H]

    def forward_inc(self, inputs, h_in, actions_env, agent_pos, agent_orientation, reward, clean_num, apple_den,
                    learning_mode=False):
        n, H = self.n_agents, self.hidden
        act = actions_env.reshape(-1, n, self.n_actions).to(inputs.dtype)
        B = act.shape[0]
        x = th.cat([inputs.reshape(B, n, self.input_shape), act], dim=-1).transpose(0, 1)   # [n, B, in + A]
        h = h_in.reshape(B, n, H).transpose(0, 1)
        x = F.leaky_relu(th.baddbmm(self._b("fc1_inc_b"), x, self._w("fc1_inc_w")))
        h_out = self._gru("inc", x, h)                                               # [n, B, H]
        # per ordered pair (i -> j): [h_i | onehot(a_j), pos_j, orient_j, r_j, clean_j, apple_den_j] (homophily_agent.py:194-201)
        other = th.cat([act, agent_pos.reshape(B, n, 2), agent_orientation.reshape(B, n, 2), reward.reshape(B, n, 1),
                        clean_num.reshape(B, n, 1), apple_den.reshape(B, n, 1)], dim=-1)     # [B, n(j), E]
        E = other.shape[-1]
        cat = th.cat([h_out.unsqueeze(2).expand(n, B, n, H), other.unsqueeze(0).expand(n, B, n, E)], dim=-1)   # [n(i), B, n(j), H+E]
        cat = cat.reshape(n, B * n, H + E)
        a = th.baddbmm(self._b("fc2_inc_b"), cat, self._w("fc2_inc_w"))             # [n, B*n, 3]
        v = th.baddbmm(self._b("fc2_inc_v_b"), cat, self._w("fc2_inc_v_w"))
        q = (v + a - a.mean(dim=-1, keepdim=True)).reshape(n, B, n, -1).transpose(0, 1)   # [B, n(i), n(j), 3]
        return q, h_out.transpose(0, 1).unsqueeze(2), {}
