"""Diagnostic: the per-agent affine layers alone (ssd_bias_bmm_fwd / _bwd through the C-ABI, preallocated outputs) at the train
step's shapes; HIP events around 200 back-to-back launches (the launch floor is inside the figure, as in the captured step; where
the figure is ~5 us or less the host's call rate is what is measured)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from homophily_marl_amd import abi  # noqa: E402

SHAPES = [(5, 1616, 73, 64, "fc1 env"), (5, 1616, 82, 64, "fc1 inc"), (10, 1616, 64, 192, "gi projections"), (5, 1616, 64, 9, "fc2 env"),
          (5, 1616, 64, 1, "fc2 env v"), (5, 8080, 80, 3, "fc2 inc"), (5, 8080, 80, 1, "fc2 inc v"), (1, 8080, 1014, 32, "encoder Linear"),
          (20, 1600, 64, 192, "dW_h")]
lib = abi.load_library()


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    th.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


g = th.Generator(device="cuda").manual_seed(0)
st = th.cuda.current_stream().cuda_stream
for n, R, I, O, name in SHAPES:
    x = th.randn(n, R, I, generator=g, device="cuda")
    w = th.randn(n, I, O, generator=g, device="cuda") * 0.2
    b = th.randn(n, 1, O, generator=g, device="cuda") * 0.1
    gout = th.randn(n, R, O, generator=g, device="cuda")
    y, dx, dw, db = th.empty_like(gout), th.empty_like(x), th.empty_like(w), th.empty_like(b)
    f = timed(lambda: lib.ssd_bias_bmm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n, R, I, O, st))
    bw = timed(lambda: lib.ssd_bias_bmm_bwd(gout.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), None, n, R, I, O, st))
    wo = timed(lambda: lib.ssd_bias_bmm_bwd(gout.data_ptr(), x.data_ptr(), None, None, dw.data_ptr(), None, None, n, R, I, O, st))
    print("%-16s n %2d R %5d I %4d O %3d   forward %6.1f us   backward %6.1f us   dw only %6.1f us" % (name, n, R, I, O, f, bw, wo), flush=True)
