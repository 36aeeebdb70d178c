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
import os
from collections.abc import Mapping
from types import SimpleNamespace

import torch as th
import torch.distributed as dist
from torch.optim import Adam

from .. import ops
from ..components.episode_buffer import EpisodeBatch
from ..controllers import REGISTRY as mac_REGISTRY

NEG = -9999999


class _FusedLogs(Mapping):
    """The nine logged scalars of a train step (homophily_learner.py:228-246) as a read-only mapping over the loss kernel's sums:
    a quotient is formed when it is read (the logger reads every learner_log_interval steps), not as ~14 scalar launches per step.
    Inside a captured train step the mapping keeps reading the replayed buffers."""
    KEYS = ("incentives_to_cleanup_per", "incentives_to_harvest_per", "value_give_mean", "value_receive_mean", "q_env_taken_mean",
            "q_inc_taken_mean", "loss_value_env", "loss_value_inc", "loss_sim")

    def __init__(self, sums, dens, rows, n):
        self.sums, self.dens, self.rows, self.n = sums, dens, rows, n

    def __getitem__(self, k):
        s, d, rows = self.sums, self.dens, self.rows
        with th.no_grad():
            return {"incentives_to_cleanup_per": lambda: s[9] / (s[10] + 1e-6), "incentives_to_harvest_per": lambda: s[11] / (s[12] + 1e-6),
                    "value_give_mean": lambda: s[7] / rows, "value_receive_mean": lambda: s[8] / rows,
                    "q_env_taken_mean": lambda: s[5] / rows, "q_inc_taken_mean": lambda: s[6] / (rows * self.n),
                    "loss_value_env": lambda: s[2] / d[0], "loss_value_inc": lambda: s[3] / d[0], "loss_sim": lambda: s[4] / (1 + d[1])}[k]()

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)

    def host_items(self, extra=()):
        """(key, python float) of all nine scalars (+ extra (key, 0-d tensor) pairs) from ONE device -> host copy: the sums and the
        denominators travel once and the quotients are formed on the host in f32 as above -- twelve .item() calls were twelve
        synchronisations (0.4 ms of an otherwise idle GPU per log interval)."""
        with th.no_grad():
            flat = th.cat([self.sums.detach().float().reshape(-1), self.dens.detach().float().reshape(-1)] + [v.detach().float().reshape(1) for _, v in extra])
        host = flat.cpu()
        ns = self.sums.numel()
        host_logs = _FusedLogs(host[:ns], host[ns:ns + self.dens.numel()], self.rows, self.n)
        out = [(k, float(host[ns + self.dens.numel() + i])) for i, (k, _) in enumerate(extra)]
        return out + [(k, float(host_logs[k])) for k in self.KEYS]

    def synced(self):
        """Under data parallelism the kernel sums are this rank's share while the denominators are global: the logger reads the
        sums added over the group (one small all-reduce per log interval, issued by every rank at the same train step)."""
        s = self.sums.detach().clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        return _FusedLogs(s, self.dens, self.rows * dist.get_world_size(), self.n)


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
        # train_graph: capture the step as hipGraphs (torch.cuda.CUDAGraph); needs capturable Adam (same arithmetic,
        # the step counter lives on the device)
        on_gpu = str(self.device).startswith("cuda")
        self.use_graph = bool(getattr(args, "train_graph", False)) and on_gpu
        # fused: ONE multi-tensor kernel per optimiser step instead of ~90 small launches (same Adam arithmetic; the step counters
        # then live on the device, as with capturable)
        self.fused_adam = on_gpu and bool(getattr(args, "fused_adam", True))
        self.optimiser_env = Adam(params=self.params_env, lr=args.lr_env, capturable=self.use_graph, fused=self.fused_adam or None)
        self.optimiser_inc = Adam(params=self.params_inc, lr=args.lr_inc, capturable=self.use_graph, fused=self.fused_adam or None)
        self._flat_grad = None
        self._graph = None
        self._opt_plan = None          # see _optimiser_plan (None: look again, False: tensor-op tail)
        self._static_batch = None
        self._graph_calls = 0
        # target network: a second controller with the same weights (the reference deep-copies the controller, :47)
        self.target_mac = mac_REGISTRY[args.mac](scheme, None, args)
        self.target_mac.load_state(mac)
        for p in self.target_mac.parameters():
            p.requires_grad_(False)
        self.log_stats_t = -self.args.learner_log_interval - 1
        # log_clock: the clock learner_log_interval is measured on (None: t_env, the reference's).  run.setup sets it to the
        # runner's schedule clock under schedule_unit "rollouts" (episode_limit per rollout of ALL envs), so that the vectorised loop
        # logs -- and synchronises with the host -- every learner_log_interval / episode_limit rollouts like the reference does,
        # not after every train step (one rollout of 4096 envs advances t_env by 409 600)
        self.log_clock = None
        # SSD_FORCE_DIST=1: a process group of ONE rank still issues every collective (RCCL rehearsal on a one-GPU box)
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or
                                                                               os.environ.get("SSD_FORCE_DIST") == "1")
        # profile_collectives (bench.py, N > 1): every gradient all-reduce is bracketed by a pair of events on the launch stream (the
        # collective's own stream is joined into it) and by the host clock; collective_times() reports both
        self.profile_collectives = False
        self._coll_events, self._coll_wall = [], []
        # learner_dtype: "fp32" (default: exact-f32 / f32-equivalent split products) or "bf16" -- the labelled reduced-precision variant:
        # single bf16 MFMA products in the affine layers, the recurrence and the encoder; master weights, Adam and the loss stay f32
        ld = str(getattr(args, "learner_dtype", "fp32")).lower()
        if ld not in ("fp32", "float32", "bf16", "bfloat16"):
            raise ValueError("learner_dtype must be fp32 or bf16, got %r" % (ld,))
        self.precision = 1 if ld in ("bf16", "bfloat16") else 2

    # ---- network unroll ---------------------------------------------------------------------------------------
    @staticmethod
    def unroll(mac, batch):
        """q_env [B, T, n, A], q_inc [B, T, n, n, 3] for t = 0..T-1 (the loops of homophily_learner.py:68-91)."""
        return mac.unroll(batch)

    def unroll_pair(self, batch):
        """(q_env, q_inc) of the live net and (tq_env, tq_inc) of the target net (no grad) on the same batch: the weight-independent
        preparation is shared and both recurrences (4n weight sets) run in ONE sequence launch per direction."""
        mac, tgt = self.mac, self.target_mac
        shared = mac.unroll_shared(batch)
        gi, wh, bh = mac.unroll_pre(batch, shared)
        with th.no_grad():
            gi_t, wh_t, bh_t = tgt.unroll_pre(batch, shared)
        # live env / inc, target env / inc: the four projection outputs AND the four recurrence weight images go to the sequence kernel
        # as they are (no concatenation), and the states come back as four tensors (no slicing: a slice's backward is a zero-fill + copy)
        he, hi, he_t, hi_t = ops.gru_sequence_parts(list(gi) + list(gi_t), batch.max_seq_length, batch.batch_size, list(wh) + list(wh_t), list(bh) + list(bh_t))
        q_env, q_inc = mac.agent.unroll_post(he, hi, shared["other"])
        with th.no_grad():
            tq_env, tq_inc = tgt.agent.unroll_post(he_t.detach(), hi_t.detach(), shared["other"])
        return q_env, q_inc, tq_env, tq_inc

    def _all_reduce_grad(self):
        """ONE flat fp32 collective (RCCL over xGMI) on the gradient buffer; timed when profile_collectives is set."""
        if not self.profile_collectives:
            dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
            return
        import time
        t0 = time.perf_counter()
        ev = None
        if self._flat_grad.is_cuda:
            ev = (th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
        if ev is not None:
            ev[1].record()
            self._coll_events.append(ev)
        self._coll_wall.append(time.perf_counter() - t0)

    def collective_times(self, reset=True):
        """dict(calls, device_ms_avg / max (events on the launch stream around the all-reduce), host_ms_avg (time the host spent in the
        call)) of the gradient all-reduces since the last reset; synchronises."""
        out = dict(calls=len(self._coll_wall))
        if self._coll_events:
            th.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self._coll_events]
            out.update(device_ms_avg=sum(ms) / len(ms), device_ms_max=max(ms))
        if self._coll_wall:
            out["host_ms_avg"] = 1e3 * sum(self._coll_wall) / len(self._coll_wall)
        if reset:
            self._coll_events, self._coll_wall = [], []
        return out

    def _global(self, x):
        """sum of a scalar tensor over the data-parallel group (loss denominators)."""
        if self.distributed:
            x = x.clone()
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
        return x

    # ---- one optimisation step ---------------------------------------------------------------------------------
    # Three stages so that the data-parallel collectives sit BETWEEN capturable pieces:
    #   denominators(batch)            batch data only -> global mask / sim-mask sums        [+ all-reduce of 2 scalars]
    #   forward_backward(batch, dens)  losses with global denominators, backward into the flat gradient buffer
    #                                                                                       [+ ONE all-reduce of it]
    #   clip_and_step()                clip inc, clip env, Adam inc, Adam env (homophily_learner.py:223-226)
    def _sim_inputs(self, batch):
        """windowed activity flags -> similarity mask (homophily_learner.py:184-206,214); depends on batch data only."""
        a, n, h = self.args, self.n_agents, self.sim_horizon
        rewards = batch["reward"][:, :-1] / a.reward_scale
        clean_num = (batch["clean_num"][:, :-1] > 0).float()
        cn_cum, rw_cum = th.cumsum(clean_num, dim=1), th.cumsum(rewards, dim=1)
        cn_h, rw_h = cn_cum.clone(), rw_cum.clone()
        cn_h[:, h:] -= cn_cum[:, :-h]
        rw_h[:, h:] -= rw_cum[:, :-h]
        clean_num_t, rewards_t = (cn_h > 0).float(), (rw_h > 0).float()
        which_cluster = 2 * rewards_t + clean_num_t            # exact-value clustering rule (replaces x-means, :194-203)
        is_idle = clean_num_t + rewards_t
        idle_agent = (is_idle.unsqueeze(2) * is_idle.unsqueeze(3)).unsqueeze(-1)
        similarity = (which_cluster.unsqueeze(2) == which_cluster.unsqueeze(3)).unsqueeze(-1).float() * idle_agent
        return th.relu(similarity) * self.env_sim_mask * self.inc_sim_mask * self.oth_sim_mask

    def _td_mask(self, batch):
        terminated = batch["terminated"][:, :-1].float()
        mask = batch["filled"][:, :-1].float()
        mask[:, 1:] = mask[:, 1:] * (1 - terminated[:, :-1])                   # [bs, t-1, 1]
        return mask.expand(-1, -1, self.n_agents)

    def _fused(self, batch):
        """The loss and its gradient w.r.t. the Q-values as ONE HIP launch (ssd_td_sim_loss) instead of ~100 tensor ops + autograd:
        device batches with the shipped loss flags; everything else (CPU tensors of the CPU suite / gloo rehearsal,
        consider_others_inc) keeps the tensor-op statement below, which is also what the GPU tests check the kernel against."""
        return (bool(getattr(self.args, "fused_loss", True)) and batch["reward"].is_cuda and not self.args.consider_others_inc
                and self.n_agents >= 2)

    def denominators(self, batch):
        """[mask.sum(), sim_mask.sum()] of the GLOBAL batch (all-reduced over the data-parallel group)."""
        if self._fused(batch):
            d = ops.loss_denominators(batch, self.args, self.n_actions)
        else:
            d = th.stack([self._td_mask(batch).sum(), self._sim_inputs(batch).sum()])
        if self.distributed:
            dist.all_reduce(d, op=dist.ReduceOp.SUM)
        return d

    def _bind_flat_grad(self):
        """All gradients are views into ONE persistent flat fp32 buffer: the data-parallel all-reduce is one collective on the
        buffer itself, and graph replays see stable addresses."""
        if self._flat_grad is None:
            self._flat_grad = th.zeros(sum(p.numel() for p in self.params), dtype=th.float32, device=self.params[0].device)
            off = 0
            for p in self.params:
                p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
                off += p.numel()

    def _backward(self, loss, seeds=None):
        """optimiser_{inc,env}.zero_grad() + loss.backward() (homophily_learner.py:220-222): the gradients come back as fresh tensors
        and ONE concatenation writes them into the flat buffer (accumulating into 44 pre-zeroed .grad views costs a launch each).
        seeds: `loss` is a list of tensors and seeds their gradients (the fused loss kernel's dL/dq: no loss node in the tape)."""
        self._bind_flat_grad()
        if getattr(self.args, "grad_by_cat", True):
            grads = th.autograd.grad(loss, self.params, grad_outputs=seeds)
            th.cat([g.reshape(-1) for g in grads], out=self._flat_grad)
        else:
            self._flat_grad.zero_()
            th.autograd.backward(loss, grad_tensors=seeds)

    def forward_backward(self, batch, dens=None):
        """dens None: this process's own denominators (no data parallelism: nothing to all-reduce between the stages, so the captured
        step computes them inside the graph)."""
        if dens is None:
            dens = self.denominators(batch)
        if self._fused(batch):
            return self._forward_backward_fused(batch, dens)
        return self._forward_backward_ops(batch, dens)

    def _forward_backward_fused(self, batch, dens):
        a, n = self.args, self.n_agents
        q_env, q_inc, tq_env, tq_inc = self.unroll_pair(batch)
        dq_env, dq_inc, sums = ops.td_sim_loss_grads(q_env, q_inc, tq_env, tq_inc, dens, batch, a)
        self._backward([q_env, q_inc], seeds=[dq_env, dq_inc])
        return _FusedLogs(sums, dens, float(batch.batch_size * (batch.max_seq_length - 1) * n), n)

    def _forward_backward_ops(self, batch, dens):
        a = self.args
        n = self.n_agents
        logs = {}
        rewards = batch["reward"][:, :-1] / a.reward_scale                     # [bs, t-1, n]
        actions = batch["actions"][:, :-1]                                     # [bs, t-1, n, 1]
        actions_inc = batch["actions_inc"][:, :-1]                             # [bs, t-1, n, n, 1]
        actions_inc_all = batch["actions_inc"]                                 # [bs, t, n, n, 1]
        clean_num = (batch["clean_num"][:, :-1] > 0).float()
        terminated = batch["terminated"][:, :-1].float()
        avail_actions = batch["avail_actions"]
        mask = self._td_mask(batch)
        sim_mask = self._sim_inputs(batch)

        q_env, q_inc, tq_env, tq_inc = self.unroll_pair(batch)
        with th.no_grad():
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
        target_q_env = target_q_env.masked_fill(avail_actions[:, 1:] == 0, NEG)
        other = (target_q_inc[..., 0] * recv_zero_all[:, 1:].unsqueeze(2) + target_q_inc[..., 1] * recv_pos_all[:, 1:].unsqueeze(2)
                 + target_q_inc[..., 2] * recv_neg_all[:, 1:].unsqueeze(2))
        target_next_inc = th.gather(target_q_inc, dim=-1, index=actions_inc_all[:, 1:]).squeeze(-1)
        if a.double_q:
            qe = q_env.detach().masked_fill(avail_actions == 0, NEG)
            best_env = qe[:, 1:].max(dim=-1, keepdim=True)[1]
            best_inc = q_inc.detach()[:, 1:].max(dim=-1, keepdim=True)[1]
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
        value_loss_env = ((td_env * mask) ** 2).sum() / dens[0]
        value_loss_inc = ((td_inc * mask) ** 2).sum() / dens[0]

        # similarity loss (:208-217)
        p_inc = th.softmax(q_inc, dim=-1)[:, :-1]                              # [bs, t-1, n(i), n(j), 3]
        # probability agent i assigns to the incentive action agent k actually gave to j: [bs, t-1, i, k, j]
        idx = actions_inc.squeeze(-1).unsqueeze(2).expand(-1, -1, n, -1, -1)   # [bs, t-1, (i), k, j]
        p_ikj = th.gather(p_inc.unsqueeze(3).expand(-1, -1, -1, n, -1, -1), dim=-1, index=idx.unsqueeze(-1)).squeeze(-1)
        sim_loss = (th.clamp_min(-th.log(p_ikj), a.sim_threshold) * sim_mask).sum() / (1 + dens[1])

        self._backward(value_loss_inc + value_loss_env + sim_loss * a.sim_loss_weight)

        with th.no_grad():
            q_inc_taken = th.gather(q_inc[:, :-1], dim=-1, index=actions_inc).squeeze(-1)
            logs["incentives_to_cleanup_per"] = (clean_num * receive_value).sum() / (clean_num.sum() + 1e-6)
            logs["incentives_to_harvest_per"] = (rewards * receive_value).sum() / (rewards.sum() + 1e-6)
            logs["value_give_mean"] = give_value.mean()
            logs["value_receive_mean"] = receive_value.mean()
            logs["q_env_taken_mean"] = chosen_env.mean()
            logs["q_inc_taken_mean"] = q_inc_taken.mean()
            logs["loss_value_env"] = value_loss_env.detach()
            logs["loss_value_inc"] = value_loss_inc.detach()
            logs["loss_sim"] = sim_loss.detach()
        return logs

    def _optimiser_plan(self):
        """Arguments of ssd_clip_adam_step (both clips + both Adam steps as two launches over the flat gradient buffer), or None when
        the tensor-op tail below has to run: CPU, optimiser state not created yet (the first step creates it), options the kernel does
        not implement.  The plan holds pointers into the optimisers' own state tensors (checkpoints see the same state)."""
        if self._opt_plan is not None:
            return self._opt_plan or None
        self._opt_plan = False
        a = self.args
        if not (getattr(a, "fused_optimiser", True) and self._flat_grad is not None and self._flat_grad.is_cuda):
            return None
        import ctypes as C
        from .. import abi
        opts = (self.optimiser_inc, self.optimiser_env)
        for opt in opts:
            g = opt.param_groups
            if len(g) != 1 or g[0].get("weight_decay", 0) != 0 or g[0].get("amsgrad") or g[0].get("maximize") or \
                    tuple(g[0]["betas"]) != tuple(opts[0].param_groups[0]["betas"]) or g[0]["eps"] != opts[0].param_groups[0]["eps"]:
                return None
        inc_ids, env_ids = {id(p) for p in self.params_inc}, {id(p) for p in self.params_env}
        if len(self.params) > abi.ADAM_MAX_JOBS:
            return None
        jobs = (abi.SsdAdamJob * len(self.params))()
        off = 0
        for j, p in zip(jobs, self.params):
            seg = 0 if (id(p) in inc_ids and id(p) in env_ids) else (1 if id(p) in env_ids else 2)
            if not (p.is_cuda and p.dtype == th.float32 and p.is_contiguous()) or \
                    p.grad is None or p.grad.data_ptr() != self._flat_grad.data_ptr() + 4 * off:
                self._opt_plan = None if p.grad is None else False
                return None
            for o, (opt, member) in enumerate(zip(opts, (id(p) in inc_ids, id(p) in env_ids))):
                if not member:
                    continue
                st = opt.state.get(p)
                if not st or not all(th.is_tensor(st.get(k)) and st[k].is_cuda and st[k].dtype == th.float32 and st[k].is_contiguous()
                                     for k in ("step", "exp_avg", "exp_avg_sq")):
                    self._opt_plan = None      # state is created by the first torch step: look again next time
                    return None
                j.exp_avg[o], j.exp_avg_sq[o], j.step[o] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
            j.param, j.offset, j.numel, j.segment = p.data_ptr(), off, p.numel(), seg
            off += p.numel()
        dev = self._flat_grad.device
        table = th.frombuffer(bytearray(bytes(jobs)), dtype=th.uint8).to(dev)
        args = abi.SsdClipAdamArgs()
        args.flat_grad, args.total = self._flat_grad.data_ptr(), off
        args.jobs, args.n_jobs = table.data_ptr(), len(self.params)
        partials = th.zeros((off + 1023) // 1024, 3, dtype=th.float32, device=dev)
        args.partials = partials.data_ptr()
        gi, ge = opts[0].param_groups[0], opts[1].param_groups[0]
        args.lr_inc, args.lr_env, args.beta1, args.beta2, args.eps = float(gi["lr"]), float(ge["lr"]), float(gi["betas"][0]), float(gi["betas"][1]), float(gi["eps"])
        args.clip = float(a.grad_norm_clip)
        self._opt_plan = SimpleNamespace(args=args, keep=(table, partials), lib=abi.load_library(), byref=C.byref)
        return self._opt_plan

    def clip_and_step(self):
        plan = self._optimiser_plan()
        if plan is not None:
            from .. import abi
            abi.check(plan.lib, plan.lib.ssd_clip_adam_step(plan.byref(plan.args), th.cuda.current_stream(self._flat_grad.device).cuda_stream))
            return
        a = self.args
        th.nn.utils.clip_grad_norm_(self.params_inc, a.grad_norm_clip)
        th.nn.utils.clip_grad_norm_(self.params_env, a.grad_norm_clip)
        self.optimiser_inc.step()
        self.optimiser_env.step()

    def cal_loss_and_step(self, batch):
        dens = self.denominators(batch)
        logs = self.forward_backward(batch, dens)
        if self.distributed:
            self._all_reduce_grad()                                   # ONE flat fp32 collective (RCCL over xGMI)
        self.clip_and_step()
        return logs

    # ---- hipGraph path ---------------------------------------------------------------------------------------------
    def _graph_step(self, batch):
        """Replay of the captured step on a static copy of the sampled batch.  The first two calls run eagerly (warm-up
        of allocator, hipBLASLt / MIOpen plans and Adam state), the third captures forward_backward and clip_and_step
        as two graphs with the gradient all-reduce between them."""
        self._graph_calls += 1
        # Only the fused-loss step is captured: the tensor-op loss is full of ATen multi-block reductions, which are not safe in
        # a replayed hipGraph on this stack (ops.column_sums)
        if self._graph_calls <= 2 or not self._fused(batch):
            return self.cal_loss_and_step(batch)
        if self._graph is None:
            self._static_data = {k: v.clone() for k, v in batch.data.transition_data.items()}
            self._static_batch = EpisodeBatch(batch.scheme, batch.groups, batch.batch_size, batch.max_seq_length,
                                              data=SimpleNamespace(transition_data=self._static_data, episode_data={}),
                                              device=batch.device)
            self._static_dens = th.zeros(2, device=batch.device)
            th.cuda.synchronize()
            # the step is captured on a stream of this learner's own: per-(device, stream) kernel scratch (the row-chunked affine
            # backward's partial tiles) baked into the graph then belongs to THIS graph and nothing else
            self._capture_stream = th.cuda.Stream(device=batch.device)
            ops.reserve_bmm_scratch(self._capture_stream)
            # thread_local: with an initialised process group the RCCL watchdog thread keeps polling events while we capture
            g1, g2 = th.cuda.CUDAGraph(), th.cuda.CUDAGraph()
            with th.cuda.graph(g1, stream=self._capture_stream, capture_error_mode="thread_local"):
                # data parallel: the denominators are all-reduced OUTSIDE the graph and handed in; else they are part of the graph
                self._static_logs = self.forward_backward(self._static_batch, self._static_dens if self.distributed else None)
            with th.cuda.graph(g2, pool=g1.pool(), stream=self._capture_stream, capture_error_mode="thread_local"):
                self.clip_and_step()
            self._graph = (g1, g2)
        for k, v in batch.data.transition_data.items():
            dst = self._static_data[k]
            if v.data_ptr() != dst.data_ptr() or v.shape != dst.shape:      # not sampled straight into the static batch (sample_out)
                dst.copy_(v)
        if self.distributed:
            self._static_dens.copy_(self.denominators(self._static_batch))
        self._graph[0].replay()
        if os.environ.get("SSD_GRAPH_CHECK"):      # diagnostic: the replayed gradient against an eager evaluation of the same step
            got = self._flat_grad.clone()
            self.forward_backward(self._static_batch, self._static_dens if self.distributed else None)
            ref = self._flat_grad.clone()
            err = float((got - ref).abs().max())
            self._check_n = getattr(self, "_check_n", 0) + 1
            if err > 1e-4 * max(1.0, float(ref.abs().max())) or not bool(th.isfinite(got).all()):
                print("GRAPH != EAGER at replay %d: max |diff| %.3e (|ref| max %.3e)" % (self._check_n, err, float(ref.abs().max())), flush=True)
                off = 0
                for name, prm in self.mac.agent.named_parameters():
                    a_, b_ = got[off:off + prm.numel()], ref[off:off + prm.numel()]; off += prm.numel()
                    dd = float((a_ - b_).abs().max())
                    if dd > 1e-5 or not bool(th.isfinite(a_).all()):
                        print("      %-22s max diff %.3e  |graph| %.3e |eager| %.3e" % (name, dd, float(a_.abs().max()), float(b_.abs().max())), flush=True)
                raise SystemExit(3)
            self._flat_grad.copy_(got)
        if self.distributed:
            self._all_reduce_grad()
        self._graph[1].replay()
        return self._static_logs

    def sample_out(self):
        """The static batch of the captured train step once it exists (else None): ReplayBuffer.sample(batch_size, out=...) gathers
        the sampled episodes straight into it and train() then copies nothing."""
        return self._static_batch if self._graph is not None else None

    def train(self, batch, t_env, episode_num):
        if batch["reward"].is_cuda:
            ops.set_learner_precision(self.precision)      # process-wide kernel selection; a captured step keeps what it was captured with
        logs = self._graph_step(batch) if self.use_graph else self.cal_loss_and_step(batch)
        if (episode_num - self.last_target_update_episode) / self.args.target_update_interval >= 1.0:
            self._update_targets()
            self.last_target_update_episode = episode_num
        clock = self.log_clock() if self.log_clock is not None else t_env
        if clock - self.log_stats_t >= self.args.learner_log_interval:
            means = (("clean_num_mean", batch["clean_num"][:, :-1].mean()), ("apple_den_mean", batch["apple_den"][:, :-1].mean()))
            if self.distributed and isinstance(logs, _FusedLogs):
                logs = logs.synced()
            if isinstance(logs, _FusedLogs):
                for k, v in logs.host_items(means):                      # one device -> host copy for all eleven scalars
                    self.logger.log_stat(k, v, t_env)
            else:
                for k, v in means + tuple(logs.items()):
                    self.logger.log_stat(k, v.item(), t_env)
            self.log_stats_t = clock
            self.check_numeric_status(t_env)

    def check_numeric_status(self, t_env=0):
        """The sticky range flag of the f32-equivalent split products (SSD_ERRBIT_F16_RANGE: pack kernels, rollout heads, recurrence),
        read where the host synchronises anyway (the log interval).  numeric_guard: "raise" (default) stops the run -- beyond the range
        the products carry inf / NaN silently --, "log" records the bits as a stat, "off" skips the read."""
        guard = str(getattr(self.args, "numeric_guard", "raise"))
        if guard == "off" or not self.params[0].is_cuda:
            return 0
        bits = ops.numeric_status()
        if self.logger is not None:
            self.logger.log_stat("numeric_status_bits", bits, t_env)
        if bits and guard == "raise":
            raise FloatingPointError("numeric status bits 0x%x at t_env %d: a scaled term of the two-term f16 split products left f16's "
                                     "range (SSD_ERRBIT_F16_RANGE = 0x20); the rollout / recurrence results are no longer f32-equivalent" % (bits, t_env))
        return bits

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
        # load_state_dict takes `capturable` from the SAVED param_groups (the reference's checkpoints, or ours saved with
        # train_graph off, carry False) and leaves `step` where the file had it: restore what this learner's step needs and
        # drop any captured graph (it baked the old optimiser state in)
        for opt in (self.optimiser_env, self.optimiser_inc):
            for group in opt.param_groups:
                group["capturable"], group["fused"] = self.use_graph, (self.fused_adam or None)
                if self.fused_adam:
                    group["foreach"] = None
                for p in group["params"]:
                    st = opt.state.get(p)
                    if st and "step" in st:
                        step = st["step"] if th.is_tensor(st["step"]) else th.tensor(float(st["step"]))
                        st["step"] = step.to(device=p.device if (self.use_graph or self.fused_adam) else "cpu", dtype=th.float32)
        self._graph, self._graph_calls, self._opt_plan = None, 0, None
