"""Torch-facing wrappers of the two learner-side HIP kernels (csrc/ssd_learner.hip) through the C ABI.

Device tensors always go through libssd_hip.so (load_library() raises if it is missing: no silent fallback on a GPU
box).  For CPU tensors -- the CPU test-suite and the gloo rehearsal of the data-parallel path, where no HIP device
exists -- the same arithmetic is evaluated with torch expressions; that branch is never taken by a GPU run.
"""
import torch as th

from . import abi


def _stream(t):
    return th.cuda.current_stream(t.device).cuda_stream


def build_inputs_tail(out, offset, last_actions, last_reward, last_actions_inc, pos, pos_scale, n_actions, t0):
    """Fill out[:, offset : offset + A + n + 4] with the non-visual agent-input features of
    HomophilyMAC._build_inputs (controllers/homophily_controller.py:137-184):
    onehot(last action) | onehot(agent id) | sign(last reward) | sign(#recv+ - #recv-) | pos / ||(H, W)||.
    out: f32 [B * n, stride]; last_actions i64 [B, n]; last_reward f32 [B, n]; last_actions_inc i64 [B, n, n];
    pos f32 [B, n, 2].  t0: the t == 0 branch (the three history terms are zero, their tensors may be None)."""
    B, n = pos.shape[0], pos.shape[1]
    A = n_actions
    if out.is_cuda:
        lib = abi.load_library()
        cont = lambda x: None if x is None else x.contiguous()
        la, lr, li, p = cont(last_actions), cont(last_reward), cont(last_actions_inc), pos.contiguous()
        assert out.is_contiguous() and out.dtype == th.float32
        ptr = lambda x: None if x is None else x.data_ptr()
        abi.check(lib, lib.ssd_build_inputs(B, n, A, 1 if t0 else 0, ptr(la), ptr(lr), ptr(li), p.data_ptr(), float(pos_scale),
                                            out.data_ptr(), out.shape[1], offset, _stream(out)))
        return out
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
        hs = th.empty(G, T, B, H, dtype=gi.dtype, device=gi.device)
        need = any(ctx.needs_input_grad)
        rzn = th.empty_like(gi) if need else None
        ghn = th.empty(T, G, B, H, dtype=gi.dtype, device=gi.device) if need else None
        abi.check(lib, lib.ssd_gru_seq_fwd(gi.data_ptr(), wh.data_ptr(), bh.data_ptr(), hs.data_ptr(), None if rzn is None else rzn.data_ptr(),
                                           None if ghn is None else ghn.data_ptr(), T, G, B, _stream(gi)))
        if need:
            ctx.save_for_backward(hs, rzn, ghn, wh)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        lib = abi.load_library()
        hs, rzn, ghn, wh = ctx.saved_tensors
        dhs = dhs.contiguous()
        T, G, B, H3 = rzn.shape
        tiles = (B + 15) // 16
        d_gi = th.empty_like(rzn) if B % 16 == 0 else th.zeros_like(rzn)
        d_wh = th.empty(G, tiles, H3 // 3, H3, dtype=rzn.dtype, device=rzn.device)
        d_bh = th.empty(G, tiles, H3, dtype=rzn.dtype, device=rzn.device)
        abi.check(lib, lib.ssd_gru_seq_bwd(dhs.data_ptr(), hs.data_ptr(), rzn.data_ptr(), ghn.data_ptr(), wh.data_ptr(), d_gi.data_ptr(),
                                           d_wh.data_ptr(), d_bh.data_ptr(), T, G, B, _stream(rzn)))
        return d_gi, d_wh.sum(1) if tiles > 1 else d_wh[:, 0], (d_bh.sum(1) if tiles > 1 else d_bh[:, 0]).unsqueeze(1)


def gru_sequence(gi, wh, bh):
    """hs [G, T, B, H]: the GRU states h_1..h_T for the input-side projections gi [T, G, B, 3H] from a zero initial state."""
    T, G, B, H3 = gi.shape
    if gi.is_cuda and H3 == 192:
        return _GruSeq.apply(gi, wh, bh)
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
