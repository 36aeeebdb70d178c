"""Torch-facing wrappers of the two learner-side HIP kernels (csrc/ssd_learner.hip) through the C ABI.

Device tensors always go through libssd_hip.so (load_library() raises if it is missing: no silent fallback on a GPU
box).  For CPU tensors -- the CPU test-suite and the gloo rehearsal of the data-parallel path, where no HIP device
exists -- the same arithmetic is evaluated with torch expressions; that branch is never taken by a GPU run.
"""
import ctypes as C

import torch as th

from . import abi


def _stream(t):
    return th.cuda.current_stream(t.device).cuda_stream


# strict_device_ops: raise where a DEVICE tensor would leave the HIP kernels for a tensor-op statement (shapes / dtypes an operator's
# kernel is not instantiated for).  bench.py, smoke() and the GPU suite switch it on (config key `strict_device_ops`, or
# SSD_STRICT_DEVICE_OPS=1): a silent torch branch on a GPU run is a failure there, not a fallback.
import os as _os
STRICT = _os.environ.get("SSD_STRICT_DEVICE_OPS") == "1"


def numeric_status():
    """Sticky numeric-status bits of the current device, cleared on read (ssd_numeric_status; abi.ERRBIT_F16_RANGE: a scaled value of
    a two-term f16 split product left f16's range in a pack kernel, a rollout head or the learner's recurrence).  Synchronises."""
    lib = abi.load_library()
    bits = C.c_int32(0)
    abi.check(lib, lib.ssd_numeric_status(C.byref(bits)))
    return bits.value


def reserve_bmm_scratch(stream):
    """The per-(device, stream) scratch of the row-chunked affine backward (ssd_bias_bmm_bwd) for a stream that is about to capture
    the train step: nothing can be allocated inside a capture."""
    lib = abi.load_library()
    abi.check(lib, lib.ssd_bmm_reserve_scratch(stream.cuda_stream))


def unroll_other(actions, pos, orient, reward, clean_num, apple_den, pos_scale, n_actions):
    """(other [T * B, n, A + 7], act_tm [n, T * B, A]) of a sampled batch on the device in ONE launch (ssd_unroll_other): the
    incentive head's per-receiver features [one-hot(a_j), pos_j / scale, orient_j, r_j, clean_j, apple_den_j] (homophily_agent.py:194-201)
    and the one-hot actions agent-major.  actions i64 [B, T, n], pos / orient [B, T, n, 2], reward / clean_num / apple_den [B, T, n]."""
    lib = abi.load_library()
    B, T, n = actions.shape[:3]
    c = lambda x: x.contiguous()
    actions, pos, orient, reward, clean_num, apple_den = c(actions), c(pos.float()), c(orient.float()), c(reward.float()), c(clean_num.float()), c(apple_den.float())
    other = th.empty(T * B, n, n_actions + 7, dtype=th.float32, device=actions.device)
    act_tm = th.empty(n, T * B, n_actions, dtype=th.float32, device=actions.device)
    abi.check(lib, lib.ssd_unroll_other(actions.data_ptr(), pos.data_ptr(), orient.data_ptr(), reward.data_ptr(), clean_num.data_ptr(), apple_den.data_ptr(),
                                        float(pos_scale), B, T, n, n_actions, other.data_ptr(), act_tm.data_ptr(), _stream(other)))
    return other, act_tm


LEARNER_PRECISION = 2


def set_learner_precision(precision):
    """Arithmetic of the learner's matrix products (ssd_set_learner_precision): 2 = f32 (default), 1 = the labelled bf16 variant (single
    bf16 MFMA products in the per-agent affine layers, the recurrence and the encoder; f32 accumulation / parameters / optimiser / loss).
    CPU tensors are unaffected (the tensor-op statements are f32)."""
    global LEARNER_PRECISION
    precision = int(precision)
    if precision != LEARNER_PRECISION or precision != 2:
        lib = abi.load_library()
        abi.check(lib, lib.ssd_set_learner_precision(precision))
    LEARNER_PRECISION = precision


def set_strict(on=True):
    global STRICT
    STRICT = bool(on)


def _leaving_kernels(op, t, why):
    """called on every branch that evaluates an operator with tensor ops: fine for host tensors, an error for device tensors under
    strict_device_ops."""
    if STRICT and t is not None and t.is_cuda:
        raise RuntimeError("strict_device_ops: ops.%s would take the tensor-op branch for a device tensor (%s)" % (op, why))


def build_inputs_tail(out, offset, last_actions, last_reward, last_actions_inc, pos, pos_scale, n_actions, t0, flags=None, seq_len=0):
    """Fill out[:, offset : offset + A + n + 4] with the non-visual agent-input features of
    HomophilyMAC._build_inputs (controllers/homophily_controller.py:137-184):
    onehot(last action) | onehot(agent id) | sign(last reward) | sign(#recv+ - #recv-) | pos / ||(H, W)||.
    out: f32 [B * n, stride]; last_actions i64 [B, n]; last_reward f32 [B, n]; last_actions_inc i64 [B, n, n];
    pos f32 [B, n, 2].  t0: the t == 0 branch (the three history terms are zero, their tensors may be None).
    flags (device tensors only): abi.INPUT_* bits of any other flag set -- the blocks present, in the reference's order, incl.
    everybody's last action (obs_others_last_action) and 1 - distances (obs_distance); out must be that wide.
    seq_len = T > 0 (device tensors only): B counts [episodes, T] rows and the history tensors hold every step's OWN values -- the
    kernel reads the previous timestep's row itself (no shifted copies)."""
    B, n = pos.shape[0], pos.shape[1]
    A = n_actions
    if out.is_cuda:
        lib = abi.load_library()
        cont = lambda x: None if x is None else x.contiguous()
        la, lr, li, p = cont(last_actions), cont(last_reward), cont(last_actions_inc), pos.contiguous()
        assert out.is_contiguous() and out.dtype == th.float32
        ptr = lambda x: None if x is None else x.data_ptr()
        word = (1 if t0 else 0) | (int(seq_len) << 8)
        if flags is not None:       # any _build_inputs flag set (abi.INPUT_* bits), blocks in the reference's order
            abi.check(lib, lib.ssd_build_inputs_flags(B, n, A, word, abi.INPUT_EXPLICIT | int(flags), ptr(la), ptr(lr), ptr(li),
                                                      p.data_ptr(), float(pos_scale), out.data_ptr(), out.shape[1], offset, _stream(out)))
            return out
        abi.check(lib, lib.ssd_build_inputs(B, n, A, word, ptr(la), ptr(lr), ptr(li), p.data_ptr(), float(pos_scale),
                                            out.data_ptr(), out.shape[1], offset, _stream(out)))
        return out
    assert flags is None and not seq_len, "CPU tensors: only the shipped flag set has a tensor-op statement here (HomophilyMAC.assemble_inputs has all)"
    o = out.view(B, n, -1)
    if t0:
        o[..., offset:offset + A] = 0
        o[..., offset + A + n] = 0
        o[..., offset + A + n + 1] = 0
    else:
        valid = (last_actions >= 0).unsqueeze(-1)                        # index -1 = "no previous action": all-zero one-hot
        o[..., offset:offset + A] = th.nn.functional.one_hot(last_actions.clamp(min=0), A).to(o.dtype) * valid
        o[..., offset + A + n] = th.sign(last_reward)
        m = last_actions_inc * (1 - th.eye(n, dtype=last_actions_inc.dtype, device=o.device))
        recv = (m == 1).sum(dim=1) - (m == 2).sum(dim=1)                 # giver dim summed -> per receiver
        o[..., offset + A + n + 1] = th.sign(recv.to(o.dtype))
    o[..., offset + A:offset + A + n] = th.eye(n, dtype=o.dtype, device=o.device)
    o[..., offset + A + n + 2:offset + A + n + 4] = pos / pos_scale
    return out


def incentive_transfer(actions_inc, rewards, effect_ratio, cost_ratio, incentive, seq_len):
    """Incentive reward transfer (learners/homophily_learner.py:94-115).
    actions_inc i64 [B, T, n, n] (giver dim 2, receiver dim 3), rewards f32 [B, T-1, n].
    Returns give [B,T-1,n], recv_pos / recv_neg / recv_zero [B,T,n], rewards_for_env, rewards_for_inc [B,T-1,n] (f32)."""
    B, T, n = actions_inc.shape[0], actions_inc.shape[1], actions_inc.shape[2]
    dev = actions_inc.device
    if actions_inc.is_cuda:
        lib = abi.load_library()
        a = actions_inc.contiguous()
        r = rewards.contiguous().float()
        mk = lambda t: th.empty(B, t, n, dtype=th.float32, device=dev)
        give, rp, rn, rz, re, ri = mk(T - 1), mk(T), mk(T), mk(T), mk(T - 1), mk(T - 1)
        abi.check(lib, lib.ssd_incentive_transfer(B, T, n, a.data_ptr(), r.data_ptr(), float(effect_ratio), float(cost_ratio),
                                                  float(incentive), float(seq_len), give.data_ptr(), rp.data_ptr(), rn.data_ptr(),
                                                  rz.data_ptr(), re.data_ptr(), ri.data_ptr(), _stream(a)))
        return give, rp, rn, rz, re, ri
    mask = (1 - th.eye(n, dtype=actions_inc.dtype, device=dev)).view(1, 1, n, n)
    m = actions_inc * mask
    give = (m[:, :-1] != 0).sum(dim=3).float()
    rp = (m == 1).sum(dim=2).float()
    rn = (m == 2).sum(dim=2).float()
    rz = (n - 1) - rp - rn
    rv = rp[:, :-1] - rn[:, :-1]
    re = (rewards + rv * effect_ratio * incentive) / seq_len
    ri = (rewards - give * cost_ratio * incentive) / seq_len
    return give, rp, rn, rz, re, ri


def _td_loss_args(batch, a, n_actions, partials):
    """ssd_td_loss_args over the batch's tensors (contiguous copies where a view is not); returns (args, keep-alive list)."""
    c = lambda x: x.contiguous()
    B, T1, n = batch.batch_size, batch.max_seq_length, batch["reward"].shape[-1]
    keep = dict(actions=c(batch["actions"].squeeze(-1)), actions_inc=c(batch["actions_inc"].squeeze(-1)), avail=c(batch["avail_actions"]),
                reward=c(batch["reward"].float()), clean_num=c(batch["clean_num"].float()),
                terminated=c(batch["terminated"].squeeze(-1)), filled=c(batch["filled"].squeeze(-1)))
    assert keep["avail"].dtype == th.int32 and keep["terminated"].dtype == th.uint8 and keep["filled"].dtype == th.int64
    t = abi.SsdTdLossArgs()
    t.batch, t.t_slots, t.n_agents, t.n_actions, t.sim_horizon, t.double_q = B, T1, n, n_actions, int(a.sim_horizon), int(bool(a.double_q))
    t.gamma_env, t.gamma_inc, t.reward_scale = float(a.gamma_env), float(a.gamma_inc), float(a.reward_scale)
    t.incentive_ratio, t.incentive_cost, t.incentive = float(a.incentive_ratio), float(a.incentive_cost), float(a.incentive)
    t.seq_len, t.sim_threshold, t.sim_loss_weight = float(T1), float(a.sim_threshold), float(a.sim_loss_weight)
    for k, v in keep.items():
        setattr(t, k, v.data_ptr())
    t.partials = partials.data_ptr()
    return t, keep


def loss_denominators(batch, a, n_actions):
    """[mask.sum(), sim_loss_mask.sum()] of this batch (homophily_learner.py:62-64,184-206,214) from ONE launch of ssd_td_sim_loss
    (mode 0) instead of ~30 tensor ops; device tensors only."""
    lib = abi.load_library()
    B, T, n = batch.batch_size, batch.max_seq_length - 1, batch["reward"].shape[-1]
    partials = th.empty(B * T * n, abi.TD_LOSS_PARTIALS, dtype=th.float32, device=batch["reward"].device)   # mode 0 writes columns 0, 1 of every row
    t, keep = _td_loss_args(batch, a, n_actions, partials)
    abi.check(lib, lib.ssd_td_sim_loss(C.byref(t), 0, _stream(partials)))
    return column_sums(partials)[:2]


class _TdSimLoss(th.autograd.Function):
    """loss = (sum (td_env mask)^2 + sum (td_inc mask)^2) / dens[0] + sim_loss_weight * sum_sim / (1 + dens[1]) and its gradient w.r.t.
    q_env / q_inc from one launch (csrc/ssd_learner.hip: k_td_sim_loss); returns (loss, column sums of the per-row partials)."""

    @staticmethod
    def forward(ctx, q_env, q_inc, tq_env, tq_inc, dens, batch, a):
        lib = abi.load_library()
        B, T, n = batch.batch_size, batch.max_seq_length - 1, q_env.shape[2]
        q_env, q_inc, tq_env, tq_inc = q_env.contiguous(), q_inc.contiguous(), tq_env.contiguous(), tq_inc.contiguous()
        partials = th.zeros(B * T * n, abi.TD_LOSS_PARTIALS, dtype=th.float32, device=q_env.device)
        dq_env, dq_inc = th.empty_like(q_env), th.empty_like(q_inc)
        t, keep = _td_loss_args(batch, a, q_env.shape[-1], partials)
        t.q_env, t.q_inc, t.tq_env, t.tq_inc = q_env.data_ptr(), q_inc.data_ptr(), tq_env.data_ptr(), tq_inc.data_ptr()
        dens = dens.contiguous().float()
        t.dens, t.dq_env, t.dq_inc = dens.data_ptr(), dq_env.data_ptr(), dq_inc.data_ptr()
        abi.check(lib, lib.ssd_td_sim_loss(C.byref(t), 1, _stream(q_env)))
        sums = column_sums(partials)
        loss = (sums[2] + sums[3]) / dens[0] + a.sim_loss_weight * sums[4] / (1 + dens[1])
        ctx.save_for_backward(dq_env, dq_inc)
        ctx.mark_non_differentiable(sums)
        return loss, sums

    @staticmethod
    def backward(ctx, g_loss, g_sums):
        dq_env, dq_inc = ctx.saved_tensors
        return dq_env * g_loss, dq_inc * g_loss, None, None, None, None, None


def td_sim_loss(q_env, q_inc, tq_env, tq_inc, dens, batch, a):
    return _TdSimLoss.apply(q_env, q_inc, tq_env, tq_inc, dens, batch, a)


def td_sim_loss_grads(q_env, q_inc, tq_env, tq_inc, dens, batch, a):
    """(dL/dq_env, dL/dq_inc, column sums of the per-row partials) of the same loss from the same launch, WITHOUT an autograd node:
    the learner seeds the backward pass of the two Q tensors with the gradients directly (th.autograd.grad(..., grad_outputs=...)) --
    the loss scalar is a logged quantity only (_FusedLogs forms it when it is read), so the six scalar launches that assembled it and
    the two multiplications by d loss / d loss = 1 of the autograd form are gone.  Columns 13..15 of the partials are never written or read."""
    lib = abi.load_library()
    B, T, n = batch.batch_size, batch.max_seq_length - 1, q_env.shape[2]
    q_env, q_inc, tq_env, tq_inc = q_env.detach().contiguous(), q_inc.detach().contiguous(), tq_env.contiguous(), tq_inc.contiguous()
    partials = th.empty(B * T * n, abi.TD_LOSS_PARTIALS, dtype=th.float32, device=q_env.device)
    dq_env, dq_inc = th.empty_like(q_env), th.empty_like(q_inc)
    t, keep = _td_loss_args(batch, a, q_env.shape[-1], partials)
    t.q_env, t.q_inc, t.tq_env, t.tq_inc = q_env.data_ptr(), q_inc.data_ptr(), tq_env.data_ptr(), tq_inc.data_ptr()
    dens = dens.contiguous().float()
    t.dens, t.dq_env, t.dq_inc = dens.data_ptr(), dq_env.data_ptr(), dq_inc.data_ptr()
    abi.check(lib, lib.ssd_td_sim_loss(C.byref(t), 1, _stream(q_env)))
    return dq_env, dq_inc, column_sums(partials)


def fill_blocks(entries):
    """entries: (tensor, 32-bit pattern) pairs -> ONE ssd_fill_blocks launch for device tensors (contiguous, element size 4 or 8: an i64
    -1 is the pattern 0xFFFFFFFF twice); host tensors are filled one by one."""
    dev = [(t, v) for t, v in entries if t.is_cuda]
    for t, v in entries:
        if not t.is_cuda:
            t.fill_(-1 if v == 0xFFFFFFFF else 0)
            assert v in (0, 0xFFFFFFFF)
    if not dev:
        return
    lib = abi.load_library()
    for i in range(0, len(dev), abi.FILL_BLOCKS_MAX):
        part = dev[i:i + abi.FILL_BLOCKS_MAX]
        tab = (abi.SsdBlockFill * len(part))()
        for e, (t, v) in zip(tab, part):
            assert t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0
            e.dst, e.bytes, e.value = t.data_ptr(), t.numel() * t.element_size(), v
        abi.check(lib, lib.ssd_fill_blocks(tab, len(part), _stream(part[0][0])))


def runner_stats(collective_return, equality, episode_return, acc):
    """acc f64 [4] += [sum collective_return, sum equality, sum episode_return, sum episode_return^2] (EpisodeRunner's statistics of one
    rollout, episode_runner.py:121-152): one launch on the device."""
    if acc.is_cuda and all(t.is_cuda and t.dtype == th.float32 and t.is_contiguous() for t in (collective_return, equality, episode_return)):
        lib = abi.load_library()
        abi.check(lib, lib.ssd_runner_stats(collective_return.data_ptr(), equality.data_ptr(), episode_return.data_ptr(), collective_return.numel(),
                                            episode_return.numel(), acc.data_ptr(), _stream(acc)))
        return acc
    _leaving_kernels("runner_stats", acc, "dtype / layout")
    r = episode_return.to(th.float64)
    acc += th.stack([collective_return.sum(dtype=th.float64), equality.sum(dtype=th.float64), r.sum(), (r * r).sum()])
    return acc


def column_sums(x):
    """x [R, C] -> [C] or x [G, R, C] -> [G, C]: row sums by the HIP kernel k_column_sums (one launch, deterministic, no
    cross-workgroup hand-off).  ATen's multi-block reduction kernels keep block-arrival semaphores that a memset node clears; inside a
    captured hipGraph they were observed to return ANOTHER reduction's partial sums on MI355X / ROCm 7.2 once other work ran
    between two replays (diverging / NaN training runs) -- so nothing inside the captured train step reduces over rows with them."""
    if not x.is_cuda:
        return x.sum(-2)
    lib = abi.load_library()
    x = x.contiguous()
    G = 1 if x.dim() == 2 else x.shape[0]
    R, Cc = x.shape[-2], x.shape[-1]
    out = th.empty((Cc,) if x.dim() == 2 else (G, Cc), dtype=th.float32, device=x.device)
    chunks = -(-R // abi.COLSUM_CHUNK)
    ws = th.empty(G, chunks, Cc, dtype=th.float32, device=x.device) if chunks > 1 else None
    abi.check(lib, lib.ssd_column_sums(x.data_ptr(), out.data_ptr(), G, R, Cc, None if ws is None else ws.data_ptr(), _stream(x)))
    return out


def _copy_blocks(entries, like):
    """entries: (src_ptr, dst_ptr, rows, cols, src_stride, dst_stride) -> ONE ssd_copy_blocks launch."""
    lib = abi.load_library()
    tab = (abi.SsdBlockCopy * len(entries))()
    for e, (src, dst, rows, cols, ss, ds) in zip(tab, entries):
        e.src, e.dst, e.rows, e.cols, e.src_stride, e.dst_stride = src, dst, rows, cols, ss, ds
    abi.check(lib, lib.ssd_copy_blocks(tab, len(entries), _stream(like)))


class _DuelingQ(th.autograd.Function):
    """q = v + a - mean_k a (homophily_agent.py:168-170, 203-207) from the layers' time-major rows straight into the batch layout
    [B, T, n, inner, K], one launch forward and one backward (ssd_dueling_q_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, a, v, B, T, inner):
        lib = abi.load_library()
        a, v = a.contiguous(), v.contiguous()
        n, K = a.shape[0], a.shape[-1]
        q = th.empty(B, T, n, inner, K, dtype=th.float32, device=a.device)
        abi.check(lib, lib.ssd_dueling_q_fwd(a.data_ptr(), v.data_ptr(), q.data_ptr(), n, T, B, inner, K, _stream(a)))
        ctx.dims = (n, T, B, inner, K)
        return q

    @staticmethod
    def backward(ctx, dq):
        lib = abi.load_library()
        n, T, B, inner, K = ctx.dims
        dq = dq.contiguous()
        da = th.empty(n, T * B * inner, K, dtype=th.float32, device=dq.device)
        dv = th.empty(n, T * B * inner, 1, dtype=th.float32, device=dq.device)
        abi.check(lib, lib.ssd_dueling_q_bwd(dq.data_ptr(), da.data_ptr(), dv.data_ptr(), n, T, B, inner, K, _stream(dq)))
        return da, dv, None, None, None


def dueling_q(a, v, B, T, inner):
    """a [n, T * B * inner, K] (rows (t * B + b) * inner + j), v [n, T * B * inner, 1] -> q = v + a - mean_k a as [B, T, n, K]
    (inner = 1) or [B, T, n, inner, K]."""
    n, K = a.shape[0], a.shape[-1]
    if a.is_cuda and a.dtype == th.float32 and K <= 16:
        q = _DuelingQ.apply(a, v, B, T, inner)
    else:
        _leaving_kernels("dueling_q", a, "K = %d > 16 or dtype %s" % (K, a.dtype))
        q = (v + a - a.mean(dim=-1, keepdim=True)).reshape(n, T, B, inner, K).permute(2, 1, 0, 3, 4)
    return q.reshape(B, T, n, K) if inner == 1 else q


class _DuelingHead(th.autograd.Function):
    """One dueling head of the learner's time-batched evaluation as TWO launches forward (the layer, the dueling combination) and three
    backward: w [n, in, K + 1] holds the advantage layer (columns 0..K-1) and the value layer (column K) side by side, b [n, 1, K + 1].
    other None (env head): rows = h [n, T * B, in].  other [T * B, inner, E] (incentive head): the layer input of row (tb, j) is
    [h[i, tb] | other[tb, j]] -- read from the two tensors where they are (ssd_bias_bmm2_fwd), never concatenated; backward, the
    gradient of h comes from the row-group sums the dueling backward emits (ssd_dueling_head_bwd gs -> ssd_bias_bmm_bwd_x)."""

    @staticmethod
    def forward(ctx, h, other, w, b, B, T, inner):
        lib = abi.load_library()
        h, w, b = h.contiguous(), w.contiguous(), b.contiguous()
        n, TB, H = h.shape
        K = w.shape[2] - 1
        st = _stream(h)
        y = th.empty(n, TB * inner, K + 1, dtype=th.float32, device=h.device)
        if other is None:
            assert inner == 1 and w.shape[1] == H
            abi.check(lib, lib.ssd_bias_bmm_fwd(h.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n, TB, H, K + 1, st))
        else:
            other = other.contiguous()
            E = other.shape[-1]
            assert other.shape[0] == TB and other.shape[1] == inner and w.shape[1] == H + E
            abi.check(lib, lib.ssd_bias_bmm2_fwd(h.data_ptr(), other.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n, TB * inner, H, E, K + 1,
                                                 inner, 1, st))
        q = th.empty((B, T, n, inner, K), dtype=th.float32, device=h.device)
        abi.check(lib, lib.ssd_dueling_head_fwd(y.data_ptr(), q.data_ptr(), n, T, B, inner, K, st))
        ctx.save_for_backward(h, w, *(() if other is None else (other,)))
        ctx.dims = (B, T, inner, K)
        return q

    @staticmethod
    def backward(ctx, dq):
        lib = abi.load_library()
        h, w = ctx.saved_tensors[:2]
        other = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        B, T, inner, K = ctx.dims
        n, TB, H = h.shape
        st = _stream(h)
        dq = dq.contiguous()
        need = ctx.needs_input_grad
        dy = th.empty(n, TB * inner, K + 1, dtype=th.float32, device=h.device)
        two = other is not None
        gs = th.empty(n, TB, K + 1, dtype=th.float32, device=h.device) if (two and need[0]) else None
        abi.check(lib, lib.ssd_dueling_head_bwd(dq.data_ptr(), dy.data_ptr(), None if gs is None else gs.data_ptr(), n, T, B, inner, K, st))
        dw = th.empty_like(w) if need[2] else None
        db = th.empty(n, 1, K + 1, dtype=th.float32, device=h.device) if need[3] else None
        ptr = lambda t: None if t is None else t.data_ptr()
        dh = None
        if not two:
            dh = th.empty_like(h) if need[0] else None
            abi.check(lib, lib.ssd_bias_bmm_bwd(dy.data_ptr(), h.data_ptr(), w.data_ptr(), ptr(dh), ptr(dw), ptr(db), None, n, TB, H, K + 1, st))
        else:
            E = other.shape[-1]
            if dw is not None or db is not None:
                abi.check(lib, lib.ssd_bias_bmm2_bwd_w(dy.data_ptr(), h.data_ptr(), other.data_ptr(), ptr(dw), ptr(db), n, TB * inner, H, E, K + 1, inner, 1, st))
            if need[0]:
                dh = th.empty_like(h)
                abi.check(lib, lib.ssd_bias_bmm_bwd_x(gs.data_ptr(), w.data_ptr(), dh.data_ptr(), n, TB, H, K + 1, (H + E) * (K + 1), st))
        return dh, None, dw, db, None, None, None


def dueling_head(h, other, w, b, B, T, inner):
    """q [B, T, n, K] (inner = 1, other None) or [B, T, n, inner, K] of one dueling head: h [n, T * B, H] the recurrence states (rows
    t * B + b), other [T * B, inner, E] or None, w [n, H (+ E), K + 1] = [advantage layer | value layer], b [n, 1, K + 1]."""
    n, K = h.shape[0], w.shape[2] - 1
    if h.is_cuda and h.dtype == th.float32 and K <= 15 and h.shape[-1] % 16 == 0:
        q = _DuelingHead.apply(h, other, w, b, B, T, inner)
    else:
        _leaving_kernels("dueling_head", h, "layout")
        TB = h.shape[1]
        x = h if other is None else th.cat([h.unsqueeze(2).expand(n, TB, inner, h.shape[-1]), other.unsqueeze(0).expand(n, TB, inner, other.shape[-1])],
                                          dim=-1).reshape(n, TB * inner, -1)
        y = th.baddbmm(b, x, w)
        a, v = y[..., :K], y[..., K:]
        q = (v + a - a.mean(dim=-1, keepdim=True)).reshape(n, T, B, inner, K).permute(2, 1, 0, 3, 4)
    return q.reshape(B, T, n, K) if inner == 1 else q


def _mix32p(x):
    x &= 0xFFFFFFFF
    x ^= x >> 17; x = (x * 0xed5ad4bb) & 0xFFFFFFFF; x ^= x >> 11; x = (x * 0xac4c1b51) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x31848bab) & 0xFFFFFFFF; x ^= x >> 14
    return x


def _sample_draw(seed, call, i):
    s0, s1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    x = _mix32p(s0 ^ ((call * 0x9E3779B9) & 0xFFFFFFFF))
    return _mix32p(x ^ ((s1 + i * 0x85EBCA6B) & 0xFFFFFFFF))


def sample_ids(seed, call, population, count, out):
    """out[0 .. count) (int64) = `count` distinct indices drawn uniformly from [0, population): the replay buffer's
    np.random.choice(population, count, replace=False) (episode_buffer.py:240-244) as a counter generator keyed by (seed, call).
    Device tensor: one launch (ssd_sample_ids), nothing crosses the host boundary.  Host tensor: the same arithmetic restated
    (Floyd's subset sampling, then Fisher-Yates over the picks), so both give the same ids."""
    assert out.dtype == th.long and out.numel() >= count and 1 <= count <= population
    if out.is_cuda:
        lib = abi.load_library()
        abi.check(lib, lib.ssd_sample_ids(seed & (2 ** 64 - 1), call & 0xFFFFFFFF, population, count, out.data_ptr(), _stream(out)))
        return out
    ids = []
    for i in range(count):
        j = population - count + i
        t = (_sample_draw(seed, call & 0xFFFFFFFF, i) * (j + 1)) >> 32
        ids.append(j if t in ids else t)
    for i in range(count - 1, 0, -1):
        k = (_sample_draw(seed, call & 0xFFFFFFFF, count + i) * (i + 1)) >> 32
        ids[i], ids[k] = ids[k], ids[i]
    out[:count] = th.tensor(ids, dtype=th.long)
    return out


def gather_rows(pairs, ids):
    """dst[e] = src[ids[e]] along axis 0 for every (src, dst) pair of contiguous device tensors, as ONE launch (ssd_gather_rows).
    ids: int64 device tensor.  Returns False (nothing done) when a pair does not qualify -- the caller indexes field by field."""
    if not pairs or len(pairs) > abi.COPY_BLOCKS_MAX or not ids.is_cuda or ids.dtype != th.long:
        return False
    n = ids.numel()
    if n < 1 or n > 65535:                      # the launch's grid-z limit: the caller indexes field by field
        return False
    for src, dst in pairs:
        if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype
                and dst.shape[0] == n and dst.shape[1:] == src.shape[1:] and src[0].numel() > 0):
            return False
    lib = abi.load_library()
    tab = (abi.SsdRowGather * len(pairs))()
    for e, (src, dst) in zip(tab, pairs):
        e.src, e.dst, e.row_bytes = src.data_ptr(), dst.data_ptr(), src[0].numel() * src.element_size()
    abi.check(lib, lib.ssd_gather_rows(tab, len(pairs), ids.contiguous().data_ptr(), n, _stream(ids)))
    return True


class _CatGroups(th.autograd.Function):
    """Several last-axis concatenations as ONE launch (ssd_copy_blocks), their gradients split again by ONE launch into contiguous
    per-input tensors -- what th.cat + CatBackward + the flattening of its strided gradient views do with one launch per tensor."""

    @staticmethod
    def forward(ctx, sizes, *tensors):
        ctx.sizes, ctx.shapes = sizes, [t.shape for t in tensors]
        tensors = [t.contiguous() for t in tensors]
        outs, entries, k = [], [], 0
        for cnt in sizes:
            grp = tensors[k:k + cnt]
            k += cnt
            total = sum(t.shape[-1] for t in grp)
            out = th.empty(grp[0].shape[:-1] + (total,), dtype=th.float32, device=grp[0].device)
            col = 0
            for t in grp:
                c = t.shape[-1]
                entries.append((t.data_ptr(), out.data_ptr() + 4 * col, t.numel() // c, c, c, total))
                col += c
            outs.append(out)
        _copy_blocks(entries, tensors[0])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        dev = next(g for g in grads if g is not None).device
        k = 0
        full = []
        for cnt, g in zip(ctx.sizes, grads):      # an unused output has no gradient: zeros
            if g is None:
                sh = ctx.shapes[k]
                g = th.zeros(tuple(sh[:-1]) + (sum(s_[-1] for s_ in ctx.shapes[k:k + cnt]),), dtype=th.float32, device=dev)
            full.append(g.contiguous())
            k += cnt
        grads = full
        flat = th.empty(sum(int(th.Size(sh).numel()) for sh in ctx.shapes), dtype=th.float32, device=dev)
        res, entries, k, off = [], [], 0, 0
        for cnt, g in zip(ctx.sizes, grads):
            total, col = g.shape[-1], 0
            for sh in ctx.shapes[k:k + cnt]:
                c, numel = sh[-1], int(th.Size(sh).numel())
                d = flat[off:off + numel]
                entries.append((g.data_ptr() + 4 * col, d.data_ptr(), numel // c, c, total, c))
                res.append(d.view(sh))
                col += c
                off += numel
            k += cnt
        _copy_blocks(entries, flat)
        return (None,) + tuple(res)


def cat_groups(groups):
    """[th.cat(g, dim=-1) for g in groups] for f32 tensors whose leading axes agree within a group.  On the device (at most
    abi.COPY_BLOCKS_MAX tensors): one launch forward, one backward, contiguous gradients."""
    flat = [t for g in groups for t in g]
    if (flat[0].is_cuda and len(flat) <= abi.COPY_BLOCKS_MAX and all(t.dtype == th.float32 and t.is_cuda for t in flat)
            and all(t.shape[:-1] == g[0].shape[:-1] for g in groups for t in g)):
        return list(_CatGroups.apply(tuple(len(g) for g in groups), *flat))
    _leaving_kernels("cat_groups", flat[0], "%d tensors / dtypes / leading axes" % len(flat))
    return [th.cat(list(g), dim=-1) for g in groups]


def _bmm_kernel_ok(x, w, b):
    return (x.is_cuda and x.dtype == th.float32 and w.dtype == th.float32 and x.dim() == 3 and w.dim() == 3)


def _bias_bmm_fwd(x, w, b, leaky=False):
    lib = abi.load_library()
    x, w, b = x.contiguous(), w.contiguous(), b.contiguous()
    n, R, I = x.shape
    O = w.shape[2]
    y = th.empty(n, R, O, dtype=th.float32, device=x.device)
    fn = lib.ssd_bias_bmm_leaky_fwd if leaky else lib.ssd_bias_bmm_fwd
    abi.check(lib, fn(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n, R, I, O, _stream(x)))
    return y


class _BiasBmm(th.autograd.Function):
    """baddbmm(b [n, 1, O], x [n, R, I], w [n, I, O]) on the device: one launch forward (ssd_bias_bmm_fwd) and ONE launch for the three
    gradients (ssd_bias_bmm_bwd: dx, dw and the bias gradient as column sums -- deterministic, no ATen multi-block reduction);
    csrc/ssd_bmm.hip.  leaky: the layer is followed by nn.LeakyReLU() (fc1 of both heads): applied in the forward kernel's epilogue
    and, backward, to the gradient as it is loaded (ssd_bias_bmm_leaky_fwd / _bwd) -- no elementwise launch in either direction."""

    @staticmethod
    def forward(ctx, x, w, b, leaky):
        x, w = x.contiguous(), w.contiguous()
        y = _bias_bmm_fwd(x, w, b, leaky)
        ctx.leaky = leaky
        ctx.save_for_backward(x, w, *((y,) if leaky else ()))
        return y

    @staticmethod
    def backward(ctx, g):
        lib = abi.load_library()
        x, w = ctx.saved_tensors[:2]
        g = g.contiguous()
        n, R, I = x.shape
        O = w.shape[2]
        need = ctx.needs_input_grad
        dx = th.empty_like(x) if need[0] else None
        dw = th.empty_like(w) if need[1] else None
        db = th.empty(n, 1, O, dtype=th.float32, device=x.device) if need[2] else None
        ptr = lambda t: None if t is None else t.data_ptr()
        if ctx.leaky:
            y = ctx.saved_tensors[2]
            abi.check(lib, lib.ssd_bias_bmm_leaky_bwd(g.data_ptr(), y.data_ptr(), x.data_ptr(), w.data_ptr(), ptr(dx), ptr(dw), ptr(db), None, n, R, I, O, _stream(x)))
        else:
            abi.check(lib, lib.ssd_bias_bmm_bwd(g.data_ptr(), x.data_ptr(), w.data_ptr(), ptr(dx), ptr(dw), ptr(db), None, n, R, I, O, _stream(x)))
        return dx, dw, db, None


def bias_bmm(x, w, b, leaky=False):
    """x @ w + b for per-agent weights: x [n, R, I], w [n, I, O], b [n, 1, O]; leaky: followed by LeakyReLU (slope 0.01)."""
    if _bmm_kernel_ok(x, w, b):
        if th.is_grad_enabled() and (w.requires_grad or b.requires_grad or x.requires_grad):
            return _BiasBmm.apply(x, w, b, leaky)
        return _bias_bmm_fwd(x, w, b, leaky)
    _leaving_kernels("bias_bmm", x, "dtype / rank")
    y = th.baddbmm(b, x, w)
    return th.nn.functional.leaky_relu(y) if leaky else y


class _BiasLinear(th.autograd.Function):
    """F.linear(x [R, I], w [O, I], b [O]) with the bias gradient from ops.column_sums."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return th.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        return (g @ w if ctx.needs_input_grad[0] else None, g.t() @ x if ctx.needs_input_grad[1] else None,
                column_sums(g) if ctx.needs_input_grad[2] else None)


class _ChannelBias(th.autograd.Function):
    """y [R, C, H, W] + b [C] (the bias of a convolution) with the bias gradient from ops.column_sums over the R rows followed by a
    short per-channel sum."""

    @staticmethod
    def forward(ctx, y, b):
        return y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        R, C = g.shape[0], g.shape[1]
        # rows first (k_column_sums), then the H x W positions of a channel as the rows of a second k_column_sums: no ATen reduction
        db = column_sums(column_sums(g.reshape(R, -1)).view(C, -1).t().contiguous()) if ctx.needs_input_grad[1] else None
        return g, db


def channel_bias(y, b):
    return _ChannelBias.apply(y, b)


def bias_linear(x, w, b):
    if x.is_cuda and (w.requires_grad or b.requires_grad or x.requires_grad):
        return _BiasLinear.apply(x, w, b)
    return th.nn.functional.linear(x, w, b)


def expand_codes(codes):
    """u8 class codes [..., V, V] of the simplified palette (0 nothing, 1 apple, 2 waste, 3 wall / agent; include/ssd_hip.h
    SSD_OBS_CODE) -> f32 [..., 3, V, V]: waste = R, apple = G, wall / agent = B at 255/256 -- the observation the env emits in
    f32 form (map_env.py:945, cleanup.py:96-105)."""
    c = codes.unsqueeze(-3)
    return th.cat([c == 2, c == 1, c == 3], dim=-3).float() * (255.0 / 256.0)


def encode_codes_supported(codes):
    return codes.is_cuda and codes.dtype == th.uint8 and codes.shape[-1] == codes.shape[-2] and codes.shape[-1] in (15, 31)


class _EncodeCodes(th.autograd.Function):
    """rgb_preprocess (Conv2d 3->6 3x3 + LeakyReLU + Flatten + Linear -> 32 + LeakyReLU; homophily_agent.py:20-27,213-214) of windows
    given as u8 class codes [R, V, V], forward on the matrix cores (ssd_policy_encode: the rollout's encoder kernel, f32-equivalent
    products): 2 launches (weight fragments, encoder) instead of the expansion to f32 planes + MIOpen convolution + GEMM + 4
    elementwise kernels.  With gradients the kernel also emits LeakyReLU(conv) [R, 6, O, O]; the backward is the reference's
    arithmetic on it: two GEMMs for the Linear, the convolution's weight gradient (MIOpen) on the re-expanded planes, the bias
    gradients by ops.column_sums."""

    @staticmethod
    def forward(ctx, codes, conv_w, conv_b, lin_w, lin_b):
        lib = abi.load_library()
        codes = codes.contiguous()
        R, V = codes.shape[0], codes.shape[-1]
        O = V - 2
        dev = codes.device
        st = _stream(codes)
        cw, lw = conv_w.detach().contiguous(), lin_w.detach().contiguous()
        cb, lb = conv_b.detach().contiguous(), lin_b.detach().contiguous()
        prec = LEARNER_PRECISION
        need = any(ctx.needs_input_grad[1:])
        # the training forward also emits LeakyReLU(conv) and runs on the Toeplitz images; a forward without gradients (the target net)
        # takes the class-LUT layout: the conv as a table sum, a third of the matrix-core work
        layout = abi.ENCODE_LAYOUT_TOEPLITZ if need else abi.ENCODE_LAYOUT_LUT
        cbytes, lbytes = abi.encode_frag_bytes(V, prec, layout)
        cf, lf = th.empty(cbytes, dtype=th.uint8, device=dev), th.empty(lbytes, dtype=th.uint8, device=dev)
        pack = lib.ssd_policy_pack_encoder if need else lib.ssd_policy_pack_encoder_lut
        abi.check(lib, pack(cw.data_ptr(), cb.data_ptr(), lw.data_ptr(), V, prec, cf.data_ptr(), lf.data_ptr(), st))
        act = th.empty(R, 6, O, O, dtype=th.float32, device=dev) if need else None
        bands = abi.encode_bands(V)
        ea = abi.SsdPolicyEncodeArgs()
        ea.codes, ea.code_bytes = codes.data_ptr(), codes.numel()
        ea.env_stride, ea.slot_stride, ea.agent_stride, ea.slot_t = V * V, 0, V * V, None
        ea.rows, ea.view_edge, ea.n_agents, ea.agent_major, ea.precision = R, V, 1, 0, prec
        ea.layout = layout
        ea.alphabet = abi.CODE_CLASS
        ea.conv_frags, ea.lin_frags, ea.conv_b, ea.lin_b = cf.data_ptr(), lf.data_ptr(), cb.data_ptr(), lb.data_ptr()
        ea.act = None if act is None else act.data_ptr()
        if bands == 1:
            feat = th.empty(R, 32, dtype=th.float32, device=dev)
            ea.out, ea.out_stride = feat.data_ptr(), 32
            abi.check(lib, lib.ssd_policy_encode(C.byref(ea), st))
        else:       # 31 x 31 windows: per-band partial sums of the Linear, finished here (each output adds `bands` values)
            part = th.empty(bands, R, 32, dtype=th.float32, device=dev)
            ea.part = part.data_ptr()
            abi.check(lib, lib.ssd_policy_encode(C.byref(ea), st))
            feat = th.nn.functional.leaky_relu(column_sums(part.view(bands, R * 32)).view(R, 32) + lb)
        if need:
            ctx.save_for_backward(codes, act, feat, cw, lw)
        return feat

    @staticmethod
    def backward(ctx, d_feat):
        lib = abi.load_library()
        codes, act, feat, cw, lw = ctx.saved_tensors
        R = codes.shape[0]
        K = act[0].numel()
        st = _stream(codes)
        g2 = th.ops.aten.leaky_relu_backward(d_feat.contiguous(), feat, 0.01, True)   # dL/d(Linear output): LeakyReLU'(x) from the sign of LeakyReLU(x), one launch
        d_lin_b = column_sums(g2)
        # the two products of the Linear's backward on the per-agent-layer kernels (csrc/ssd_bmm.hip, one weight set):
        #   d_lin_w [32, K] = g2^T act   (its "dw" role: rows = K of the product, split over 16 waves per tile)
        #   d_act   [R, K]  = (g2 lin_w) * LeakyReLU'(conv)   (its "dx" role with w = lin_w^T [K, 32]; the slope from the sign of act)
        d_lin_w = th.empty(32, K, dtype=th.float32, device=codes.device)
        abi.check(lib, lib.ssd_bias_bmm_bwd(act.data_ptr(), g2.data_ptr(), None, None, d_lin_w.data_ptr(), None, None, 1, R, 32, K, st))
        d_act = th.empty_like(act)
        lwt = lw.t().contiguous()
        abi.check(lib, lib.ssd_bias_bmm_bwd(g2.data_ptr(), None, lwt.data_ptr(), d_act.data_ptr(), None, None, act.data_ptr(), 1, R, K, 32, st))
        # conv weight / bias gradient straight from the class codes (no f32 planes, no im2col): per-wave partial sums + one column sum
        P = lib.ssd_conv_wgrad_partial_rows(R)
        part = th.empty(P, 168, dtype=th.float32, device=codes.device)
        abi.check(lib, lib.ssd_conv_wgrad_codes(codes.data_ptr(), d_act.data_ptr(), part.data_ptr(), R, codes.shape[-1], st))
        tot = column_sums(part)
        return None, tot[:162].view(cw.shape), tot[162:], d_lin_w, d_lin_b


def encode_codes(codes, conv_w, conv_b, lin_w, lin_b):
    """features [R, 32] of windows given as u8 class codes [R, V, V] (V = 15 or 31, on the device): see _EncodeCodes."""
    return _EncodeCodes.apply(codes, conv_w, conv_b, lin_w, lin_b)


class _GruGates(th.autograd.Function):
    """h' = GRU gate arithmetic (homophily_agent.py:162-165,188-191) on gi = x W_i + b_i, gh = h W_h + b_h ([R, 3H], (r, z, n)
    order) as ONE forward and ONE backward HIP kernel instead of ~9 + ~20 pointwise launches per recurrence step."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        lib = abi.load_library()
        gi, gh, h = gi.contiguous(), gh.contiguous(), h.contiguous()
        R, H = h.numel() // h.shape[-1], h.shape[-1]
        h_new, rzn = th.empty_like(h), th.empty_like(gi)
        abi.check(lib, lib.ssd_gru_gates_fwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), h_new.data_ptr(), rzn.data_ptr(), R, H, _stream(h)))
        ctx.save_for_backward(rzn, gh, h)
        return h_new

    @staticmethod
    def backward(ctx, dh):
        lib = abi.load_library()
        rzn, gh, h = ctx.saved_tensors
        dh = dh.contiguous()
        R, H = h.numel() // h.shape[-1], h.shape[-1]
        d_gi, d_gh, dh_prev = th.empty_like(rzn), th.empty_like(rzn), th.empty_like(h)
        abi.check(lib, lib.ssd_gru_gates_bwd(dh.data_ptr(), rzn.data_ptr(), gh.data_ptr(), h.data_ptr(), d_gi.data_ptr(), d_gh.data_ptr(),
                                             dh_prev.data_ptr(), R, H, _stream(h)))
        return d_gi, d_gh, dh_prev


class _GruSeq(th.autograd.Function):
    """The whole recurrence h_t = GRU(gi_t, h_{t-1}) for t = 0..T-1 from h = 0 as one forward and one backward launch
    (csrc/ssd_gru_seq.hip).  gi [T, G, B, 3H], wh [G, H, 3H], bh [G, 1, 3H] -> hs [G, T, B, H]."""

    @staticmethod
    def forward(ctx, gi, wh, bh):
        lib = abi.load_library()
        gi, wh, bh = gi.contiguous(), wh.contiguous(), bh.contiguous()
        T, G, B, H3 = gi.shape
        H = H3 // 3
        Bp = (B + 15) // 16 * 16                    # the kernels walk whole 16-row tiles: ragged batches are padded with zero rows
        if Bp != B:
            gi = th.nn.functional.pad(gi, (0, 0, 0, Bp - B))
        hs = th.empty(G, T, Bp, H, dtype=gi.dtype, device=gi.device)
        need = any(ctx.needs_input_grad)
        rzn = th.empty_like(gi) if need else None
        ghn = th.empty(T, G, Bp, H, dtype=gi.dtype, device=gi.device) if need else None
        abi.check(lib, lib.ssd_gru_seq_fwd(gi.data_ptr(), wh.data_ptr(), bh.data_ptr(), hs.data_ptr(), None if rzn is None else rzn.data_ptr(),
                                           None if ghn is None else ghn.data_ptr(), T, G, Bp, _stream(gi)))
        if need:
            ctx.save_for_backward(hs, rzn, ghn, wh)
            ctx.B = B
        return hs if Bp == B else hs[:, :, :B].contiguous()

    @staticmethod
    def backward(ctx, dhs):
        lib = abi.load_library()
        hs, rzn, ghn, wh = ctx.saved_tensors
        T, G, Bp, H3 = rzn.shape
        B = ctx.B
        dhs = dhs.contiguous() if Bp == B else th.nn.functional.pad(dhs, (0, 0, 0, Bp - B))      # padded rows: zero gradient
        tiles = Bp // 16
        d_gi = th.empty_like(rzn)
        dgh = th.empty(G, T, Bp, H3, dtype=rzn.dtype, device=rzn.device)      # workspace: dL/dgh_t, the operand of the W_h gradient
        d_wh = th.empty(G, H3 // 3, H3, dtype=rzn.dtype, device=rzn.device)
        d_bh = th.empty(G, tiles, H3, dtype=rzn.dtype, device=rzn.device)
        abi.check(lib, lib.ssd_gru_seq_bwd(dhs.data_ptr(), hs.data_ptr(), rzn.data_ptr(), ghn.data_ptr(), wh.data_ptr(), d_gi.data_ptr(),
                                           dgh.data_ptr(), d_wh.data_ptr(), d_bh.data_ptr(), T, G, Bp, _stream(rzn)))
        if Bp != B:
            d_gi = d_gi[:, :, :B].contiguous()
        return d_gi, d_wh, (column_sums(d_bh) if tiles > 1 else d_bh[:, 0]).unsqueeze(1)      # per-tile bias sums: k_column_sums (no ATen reduction)


class _GruSeqParts(th.autograd.Function):
    """_GruSeq with the input-side projections given as separately allocated set-major parts [sets, T * B, 3H] (the per-agent affine
    layers' outputs as they are: ssd_gru_seq_fwd_parts / _bwd_parts) -- no concatenation / transpose copies, and the gradients come back
    per part, contiguous.  The recurrence weights come as n_w separately allocated parts too (wh [sets, H, 3H], bh [sets, 1, 3H]: the
    live net's and the target net's images as they are).  B is a multiple of 16 here (gru_sequence_parts falls back otherwise)."""

    @staticmethod
    def forward(ctx, T, B, n_w, *tensors):
        lib = abi.load_library()
        whs, bhs = [t.contiguous() for t in tensors[:n_w]], [t.contiguous() for t in tensors[n_w:2 * n_w]]
        parts = [p.contiguous() for p in tensors[2 * n_w:]]
        G, H3 = sum(p.shape[0] for p in parts), parts[0].shape[-1]
        H = H3 // 3
        dev = parts[0].device
        hs = th.empty(G, T, B, H, dtype=th.float32, device=dev)
        need = any(ctx.needs_input_grad)
        rzn = th.empty(T, G, B, H3, dtype=th.float32, device=dev) if need else None
        ghn = th.empty(T, G, B, H, dtype=th.float32, device=dev) if need else None
        ptrs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        wptrs = (C.c_void_p * n_w)(*[w.data_ptr() for w in whs])
        bptrs = (C.c_void_p * n_w)(*[b.data_ptr() for b in bhs])
        abi.check(lib, lib.ssd_gru_seq_fwd_parts(ptrs, len(parts), wptrs, bptrs, n_w, hs.data_ptr(), None if rzn is None else rzn.data_ptr(),
                                                 None if ghn is None else ghn.data_ptr(), T, G, B, _stream(hs)))
        if need:
            ctx.save_for_backward(hs, rzn, ghn, *whs)
            ctx.shapes = [p.shape for p in parts]
            ctx.n_w = n_w
        ctx.set_materialize_grads(False)       # states without a gradient (the target net's) arrive as None, not as zero tensors
        spp = parts[0].shape[0]
        # one output per projection part: views of the one state buffer (slicing a single output would cost a zero-fill + copy + add per
        # slice in the backward pass)
        return tuple(hs[k * spp:(k + 1) * spp] for k in range(len(parts)))

    @staticmethod
    def backward(ctx, *dhs_parts):
        lib = abi.load_library()
        hs, rzn, ghn = ctx.saved_tensors[:3]
        whs = ctx.saved_tensors[3:]
        n_w = ctx.n_w
        T, G, B, H3 = rzn.shape
        n_parts = len(ctx.shapes)
        spp = G // n_parts
        need = ctx.needs_input_grad
        # the sets whose states carry a gradient form a prefix (the live net's parts come first); the kernel walks only those
        last = max([k for k, d in enumerate(dhs_parts) if d is not None], default=-1)
        if last < 0:
            return (None,) * len(need)
        Gn = (last + 1) * spp
        dhs = [(d.contiguous() if d is not None else th.zeros(spp, T, B, H3 // 3, dtype=th.float32, device=rzn.device)) for d in dhs_parts[:last + 1]]
        d_parts = [th.empty(sh, dtype=th.float32, device=rzn.device) for sh in ctx.shapes[:last + 1]]
        dgh = th.empty(Gn, T, B, H3, dtype=th.float32, device=rzn.device)
        d_wh = th.empty(Gn, H3 // 3, H3, dtype=th.float32, device=rzn.device)
        tiles = B // 16
        d_bh = th.empty(Gn, tiles, H3, dtype=th.float32, device=rzn.device)
        pad = lambda xs: [x.data_ptr() for x in xs] + [xs[0].data_ptr()] * (n_parts - len(xs))       # parts past Gn: never touched
        ptrs = (C.c_void_p * n_parts)(*pad(d_parts))
        dptrs = (C.c_void_p * n_parts)(*pad(dhs))
        wptrs = (C.c_void_p * n_w)(*[w.data_ptr() for w in whs])
        abi.check(lib, lib.ssd_gru_seq_bwd_parts(dptrs, hs.data_ptr(), rzn.data_ptr(), ghn.data_ptr(), wptrs, n_w, ptrs, n_parts,
                                                 dgh.data_ptr(), d_wh.data_ptr(), d_bh.data_ptr(), T, G, Gn, B, _stream(rzn)))
        d_bh_all = (column_sums(d_bh) if tiles > 1 else d_bh[:, 0]).unsqueeze(1)                    # [Gn, 1, 3H]
        spw = G // n_w
        have = lambda k: (k + 1) * spw <= Gn                                                         # weight part k lies inside the walked sets
        g_wh = tuple(d_wh[k * spw:(k + 1) * spw] if (need[3 + k] and have(k)) else None for k in range(n_w))       # views of the one output: no copies
        g_bh = tuple(d_bh_all[k * spw:(k + 1) * spw] if (need[3 + n_w + k] and have(k)) else None for k in range(n_w))
        g_parts = tuple(d_parts[k] if (k <= last and need[3 + 2 * n_w + k]) else None for k in range(n_parts))
        return (None, None, None) + g_wh + g_bh + g_parts


def gru_sequence_parts(parts, T, B, wh, bh):
    """gru_sequence for projections held as equally sized set-major parts [sets, T * B, 3H] (rows t * B + b); wh [G, H, 3H], bh [G, 1, 3H]
    over all sets in part order -- each either ONE tensor or a list of 1..4 equally sized parts over the sets (no concatenation).
    Returns the states as a LIST with one tensor [sets, T, B, H] per projection part.  A weight part whose sets' states carry no
    gradient gets none (the learner puts the target net's parts last).  """
    H3 = parts[0].shape[-1]
    whs, bhs = (list(wh) if isinstance(wh, (list, tuple)) else [wh]), (list(bh) if isinstance(bh, (list, tuple)) else [bh])
    G = sum(p.shape[0] for p in parts)
    if (parts[0].is_cuda and H3 == 192 and B % 16 == 0 and 1 <= len(parts) <= 4 and all(p.shape == parts[0].shape and p.dtype == th.float32 for p in parts)
            and len(whs) == len(bhs) and 1 <= len(whs) <= 4 and all(w.shape == whs[0].shape for w in whs) and whs[0].shape[0] * len(whs) == G):
        return list(_GruSeqParts.apply(T, B, len(whs), *whs, *bhs, *parts))
    wh, bh = (whs[0] if len(whs) == 1 else th.cat(whs, dim=0)), (bhs[0] if len(bhs) == 1 else th.cat(bhs, dim=0))
    if H3 != 192 or parts[0].dtype != th.float32:
        _leaving_kernels("gru_sequence_parts", parts[0], "hidden size %d / dtype" % (H3 // 3))
    # ragged batches (B % 16) and other part counts: the time-major launch (it pads the batch itself); still the HIP recurrence
    gi = th.cat([p.reshape(p.shape[0], T, B, H3) for p in parts], dim=0).transpose(0, 1).contiguous()      # [T, G, B, 3H]
    hs = gru_sequence(gi, wh, bh)
    spp = parts[0].shape[0]
    return [hs[k * spp:(k + 1) * spp] for k in range(len(parts))]


def gru_sequence(gi, wh, bh):
    """hs [G, T, B, H]: the GRU states h_1..h_T for the input-side projections gi [T, G, B, 3H] from a zero initial state."""
    T, G, B, H3 = gi.shape
    if gi.is_cuda and H3 == 192:
        return _GruSeq.apply(gi, wh, bh)
    _leaving_kernels("gru_sequence", gi, "hidden size %d: the sequence kernels are instantiated for 64" % (H3 // 3))
    h = gi.new_zeros(G, B, H3 // 3)
    hs = []
    for t in range(T):
        h = gru_gates(gi[t], th.baddbmm(bh, h, wh), h)
        hs.append(h)
    return th.stack(hs, dim=1)


def gru_gates(gi, gh, h):
    """h' from the two projections and the previous state; [..., 3H], [..., 3H], [..., H] -> [..., H]."""
    if gi.is_cuda:
        return _GruGates.apply(gi, gh, h)
    H = h.shape[-1]
    r = th.sigmoid(gi[..., :H] + gh[..., :H])
    z = th.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    cand = th.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1 - z) * cand + z * h
