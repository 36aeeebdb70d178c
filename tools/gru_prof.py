"""Diagnostic: the learner's recurrence kernels alone (T = 101, 20 weight sets, 16 rows: the train step's shape), forward with and
without the training outputs, and backward; HIP events around back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from homophily_marl_amd import ops  # noqa: E402

T, G, B, H = 101, int(os.environ.get("G", 20)), int(os.environ.get("B", 16)), 64
g = th.Generator(device="cuda").manual_seed(0)
gi = th.randn(T, G, B, 3 * H, generator=g, device="cuda") * 0.5
wh = th.randn(G, H, 3 * H, generator=g, device="cuda") * 0.1
bh = th.randn(G, 1, 3 * H, generator=g, device="cuda") * 0.1


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    th.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


with th.no_grad():
    print("forward, no training outputs: %.1f us" % timed(lambda: ops.gru_sequence(gi, wh, bh)))
gi.requires_grad_(); wh.requires_grad_(); bh.requires_grad_()
print("forward, training outputs:    %.1f us" % timed(lambda: ops.gru_sequence(gi, wh, bh)))
hs = ops.gru_sequence(gi, wh, bh)
w = th.randn_like(hs)
print("backward:                     %.1f us" % timed(lambda: th.autograd.grad((hs * w).sum(), [gi, wh, bh], retain_graph=True)))
