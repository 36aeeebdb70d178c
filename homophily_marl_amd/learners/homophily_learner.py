"""HomophilyLearner: independent double-Q learning for the env head and the incentive head, incentive reward transfer
and the similarity ("homophily") loss.  Surface and arithmetic of the reference src/learners/homophily_learner.py:11-288
(train / cal_loss_and_step / cuda / save_models / load_models; two Adam optimisers that both own the encoder).

Differences in HOW, not WHAT:
  * the conv encoder is evaluated once for all T timesteps, only the GRU recurrences are stepped;
  * the incentive transfer (:94-115) is the HIP kernel ssd_incentive_transfer;
  * x-means clustering (pyclustering, absent here and unpinned by the reference: SURVEY.md 8(c)) is replaced by the
    documented exact-value rule cluster = 2 * rewards_t + clean_num_t, which also removes the GPU->CPU sync of :194;
  * data parallel: loss denominators and gradients are all-reduced over the process group (RCCL) before clipping.
"""
import torch as th
import torch.distributed as dist
from torch.optim import Adam

from .. import ops
from ..controllers import REGISTRY as mac_REGISTRY

NEG = -9999999


class HomophilyLearner:
    def __init__(self, mac, scheme, logger, args):
        self.args, self.mac, self.logger = args, mac, logger
        self.device = args.device
        self.n_agents = n = args.n_agents
        self.n_actions = args.n_actions
        eye = th.eye(n)
        self.inc_mask = (1 - eye).reshape(1, 1, n, n).to(self.device)
        # [bs, t-1, n(i), n(k), n(j)]: agent i, the similar agent k whose incentive action is imitated, receiver j
        self.env_sim_mask = (1 - eye).reshape(1, 1, n, n, 1).to(self.device)
        self.inc_sim_mask = (1 - eye).reshape(1, 1, n, 1, n).to(self.device)
        self.oth_sim_mask = (1 - eye).reshape(1, 1, 1, n, n).to(self.device)
        self.sim_horizon = args.sim_horizon
        self.params = list(mac.parameters())
        self.params_env = mac.parameters_env()
        self.params_inc = mac.parameters_inc()
        self.last_target_update_episode = 0
        self.optimiser_env = Adam(params=self.params_env, lr=args.lr_env)
        self.optimiser_inc = Adam(params=self.params_inc, lr=args.lr_inc)
        # target network: a second controller with the same weights (the reference deep-copies the controller, :47)
        self.target_mac = mac_REGISTRY[args.mac](scheme, None, args)
        self.target_mac.load_state(mac)
        for p in self.target_mac.parameters():
            p.requires_grad_(False)
        self.log_stats_t = -self.args.learner_log_interval - 1
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    # ---- network unroll ---------------------------------------------------------------------------------------
    @staticmethod
    def unroll(mac, batch):
        """q_env [B, T, n, A], q_inc [B, T, n, n, 3] for t = 0..T-1 (homophily_learner.py:68-76): encoder batched over
        time, recurrences stepped."""
        B, T, n = batch.batch_size, batch.max_seq_length, mac.n_agents
        a = mac.args
        obs = batch["obs"]
        if a.rgb_input:
            feat = mac.agent.rgb_preprocess(obs.reshape(B * T * n, 3, a.obs_dims[0], a.obs_dims[1]).float()).reshape(B, T, n, -1)
        else:
            feat = obs.reshape(B, T, n, -1)
        acts = batch["actions"].squeeze(-1)
        acts_inc = batch["actions_inc"].squeeze(-1)
        mac.init_hidden(B)
        q_env, q_inc = [], []
        for t in range(T):
            ft = feat[:, t].reshape(B * n, -1)
            if t == 0:
                mac.agent_inputs = mac.assemble_inputs(ft, None, None, None, batch["agent_pos"][:, 0], True)
            else:
                mac.agent_inputs = mac.assemble_inputs(ft, acts[:, t - 1], batch["reward"][:, t - 1], acts_inc[:, t - 1],
                                                       batch["agent_pos"][:, t], False)
            qe, mac.h_env, _ = mac.agent.forward_env(mac.agent_inputs, mac.h_env, True)
            q_env.append(qe.reshape(B, n, -1))
            q_inc.append(mac.forward_inc(batch, t, batch["actions"][:, t], learning_mode=True))
        return th.stack(q_env, dim=1), th.stack(q_inc, dim=1)

    def _global(self, x):
        """sum of a scalar tensor over the data-parallel group (loss denominators)."""
        if self.distributed:
            x = x.clone()
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
        return x

    # ---- one optimisation step ---------------------------------------------------------------------------------
    def cal_loss_and_step(self, batch):
        a = self.args
        n = self.n_agents
        logs = {}
        rewards = batch["reward"][:, :-1] / a.reward_scale                     # [bs, t-1, n]
        actions = batch["actions"][:, :-1]                                     # [bs, t-1, n, 1]
        actions_inc = batch["actions_inc"][:, :-1]                             # [bs, t-1, n, n, 1]
        actions_inc_all = batch["actions_inc"]                                 # [bs, t, n, n, 1]
        clean_num = (batch["clean_num"][:, :-1] > 0).float()
        terminated = batch["terminated"][:, :-1].float()
        mask = batch["filled"][:, :-1].float()
        mask[:, 1:] = mask[:, 1:] * (1 - terminated[:, :-1])                   # [bs, t-1, 1]
        avail_actions = batch["avail_actions"]

        q_env, q_inc = self.unroll(self.mac, batch)
        with th.no_grad():
            tq_env, tq_inc = self.unroll(self.target_mac, batch)
            target_q_env, target_q_inc = tq_env[:, 1:].clone(), tq_inc[:, 1:].clone()

        # incentive reward transfer (:94-115): HIP kernel
        give_value, recv_pos_all, recv_neg_all, recv_zero_all, rewards_for_env, rewards_for_inc = ops.incentive_transfer(
            actions_inc_all.squeeze(-1), rewards, a.incentive_ratio, a.incentive_cost, float(a.incentive), float(batch.max_seq_length))
        receive_positive, receive_negative, receive_zero = recv_pos_all[:, :-1], recv_neg_all[:, :-1], recv_zero_all[:, :-1]
        receive_value = receive_positive - receive_negative

        # value losses (:118-177)
        chosen_env = th.gather(q_env[:, :-1], dim=-1, index=actions)           # [bs, t-1, n, 1]
        if a.consider_others_inc:
            chosen_inc = (q_inc[:, :-1, :, :, 0] * receive_zero.unsqueeze(2) + q_inc[:, :-1, :, :, 1] * receive_positive.unsqueeze(2)
                          + q_inc[:, :-1, :, :, 2] * receive_negative.unsqueeze(2)) / (n - 1)
        else:
            chosen_inc = th.gather(q_inc[:, :-1], dim=-1, index=actions_inc).squeeze(-1)   # [bs, t-1, n, n]
        target_q_env[avail_actions[:, 1:] == 0] = NEG
        other = (target_q_inc[..., 0] * recv_zero_all[:, 1:].unsqueeze(2) + target_q_inc[..., 1] * recv_pos_all[:, 1:].unsqueeze(2)
                 + target_q_inc[..., 2] * recv_neg_all[:, 1:].unsqueeze(2))
        target_next_inc = th.gather(target_q_inc, dim=-1, index=actions_inc_all[:, 1:]).squeeze(-1)
        if a.double_q:
            qe, qi = q_env.detach().clone(), q_inc.detach()
            qe[avail_actions == 0] = NEG
            best_env = qe[:, 1:].max(dim=-1, keepdim=True)[1]
            best_inc = qi[:, 1:].max(dim=-1, keepdim=True)[1]
            tmax_env = th.gather(target_q_env, dim=-1, index=best_env)          # [bs, t-1, n, 1]
            tmax_inc_self = th.gather(target_q_inc, dim=-1, index=best_inc).squeeze(-1)
        else:
            tmax_env = target_q_env.max(dim=-1)[0]
            tmax_inc_self = target_q_inc.max(dim=-1)[0]
        tmax_inc = (tmax_inc_self + other - target_next_inc) / (n - 1) if a.consider_others_inc else tmax_inc_self

        targets_env = rewards_for_env + a.gamma_env * (1 - terminated) * tmax_env.sum(dim=-1)
        targets_inc = rewards_for_inc + a.gamma_inc * (1 - terminated) * (tmax_inc * self.inc_mask).sum(dim=-1)
        td_env = chosen_env.sum(dim=-1) - targets_env.detach()
        td_inc = (chosen_inc * self.inc_mask).sum(dim=-1) - targets_inc.detach()
        mask = mask.expand_as(td_env)
        mask_sum = self._global(mask.sum())
        value_loss_env = ((td_env * mask) ** 2).sum() / mask_sum
        value_loss_inc = ((td_inc * mask) ** 2).sum() / mask_sum

        # similarity loss (:184-217)
        h = self.sim_horizon
        cn_cum, rw_cum = th.cumsum(clean_num, dim=1), th.cumsum(rewards, dim=1)
        cn_h, rw_h = cn_cum.clone(), rw_cum.clone()
        cn_h[:, h:] -= cn_cum[:, :-h]
        rw_h[:, h:] -= rw_cum[:, :-h]
        clean_num_t, rewards_t = (cn_h > 0).float(), (rw_h > 0).float()
        which_cluster = 2 * rewards_t + clean_num_t            # exact-value clustering rule (replaces x-means, :194-203)
        is_idle = clean_num_t + rewards_t
        idle_agent = (is_idle.unsqueeze(2) * is_idle.unsqueeze(3)).unsqueeze(-1)
        similarity = (which_cluster.unsqueeze(2) == which_cluster.unsqueeze(3)).unsqueeze(-1).float() * idle_agent
        p_inc = th.softmax(q_inc, dim=-1)[:, :-1]                              # [bs, t-1, n(i), n(j), 3]
        # probability agent i assigns to the incentive action agent k actually gave to j: [bs, t-1, i, k, j]
        idx = actions_inc.squeeze(-1).unsqueeze(2).expand(-1, -1, n, -1, -1)   # [bs, t-1, (i), k, j]
        p_ikj = th.gather(p_inc.unsqueeze(3).expand(-1, -1, -1, n, -1, -1), dim=-1, index=idx.unsqueeze(-1)).squeeze(-1)
        sim_mask = th.relu(similarity.detach()) * self.env_sim_mask * self.inc_sim_mask * self.oth_sim_mask
        sim_loss = (th.clamp_min(-th.log(p_ikj), a.sim_threshold) * sim_mask).sum() / (1 + self._global(sim_mask.sum()))

        # step (:220-226)
        self.optimiser_inc.zero_grad()
        self.optimiser_env.zero_grad()
        (value_loss_inc + value_loss_env + sim_loss * a.sim_loss_weight).backward()
        if self.distributed:
            self._allreduce_grads()
        th.nn.utils.clip_grad_norm_(self.params_inc, a.grad_norm_clip)
        th.nn.utils.clip_grad_norm_(self.params_env, a.grad_norm_clip)
        self.optimiser_inc.step()
        self.optimiser_env.step()

        with th.no_grad():
            q_env_taken = chosen_env.squeeze(-1)
            q_inc_taken = th.gather(q_inc[:, :-1], dim=-1, index=actions_inc).squeeze(-1)
            logs["incentives_to_cleanup_per"] = (clean_num * receive_value).sum() / (clean_num.sum() + 1e-6)
            logs["incentives_to_harvest_per"] = (rewards * receive_value).sum() / (rewards.sum() + 1e-6)
            logs["value_give_mean"] = give_value.mean()
            logs["value_receive_mean"] = receive_value.mean()
            logs["q_env_taken_mean"] = q_env_taken.mean()
            logs["q_inc_taken_mean"] = q_inc_taken.mean()
            logs["loss_value_env"] = value_loss_env.detach()
            logs["loss_value_inc"] = value_loss_inc.detach()
            logs["loss_sim"] = sim_loss.detach()
        return logs

    def _allreduce_grads(self):
        """One all-reduce (sum) of a flat fp32 gradient buffer over RCCL / xGMI: the losses above are normalised by the
        GLOBAL denominators, so the summed shard gradients equal the gradient of the loss on the concatenated batch."""
        grads = [p.grad for p in self.params if p.grad is not None]
        flat = th.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def train(self, batch, t_env, episode_num):
        logs = self.cal_loss_and_step(batch)
        if (episode_num - self.last_target_update_episode) / self.args.target_update_interval >= 1.0:
            self._update_targets()
            self.last_target_update_episode = episode_num
        if t_env - self.log_stats_t >= self.args.learner_log_interval:
            self.logger.log_stat("clean_num_mean", batch["clean_num"][:, :-1].mean().item(), t_env)
            self.logger.log_stat("apple_den_mean", batch["apple_den"][:, :-1].mean().item(), t_env)
            for k, v in logs.items():
                self.logger.log_stat(k, v.item(), t_env)
            self.log_stats_t = t_env

    def _update_targets(self):
        self.target_mac.load_state(self.mac)
        if self.logger is not None and getattr(self.logger, "console_logger", None) is not None:
            self.logger.console_logger.info("Updated target network")

    def cuda(self):
        self.mac.cuda()
        self.target_mac.cuda()

    def save_models(self, path):
        self.mac.save_models(path)
        th.save(self.optimiser_env.state_dict(), "{}/opt_env.th".format(path))
        th.save(self.optimiser_inc.state_dict(), "{}/opt_inc.th".format(path))

    def load_models(self, path):
        self.mac.load_models(path)
        self.target_mac.load_models(path)       # reference: the target net is loaded from the same file (:281-288)
        load = lambda f: th.load("{}/{}".format(path, f), map_location=lambda storage, loc: storage, weights_only=True)
        self.optimiser_env.load_state_dict(load("opt_env.th"))
        self.optimiser_inc.load_state_dict(load("opt_inc.th"))
