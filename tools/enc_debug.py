"""Diagnostic: the encoder's response to ONE lit window cell per row (every cell x every class), saved to a file -- run once per
library build (SSD_HIP_LIB_PATH) and compare the files to see which cells a build mishandles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch as th
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_policy_mfma import _ctx
from homophily_marl_amd import abi
from homophily_marl_amd.fast_policy import FastPolicy

th.manual_seed(2)
N, n, V = 135, 5, 15
ctx = _ctx("cleanup", n, N, view=7)
fp = FastPolicy(ctx.mac, N, ctx.runner.env.avail_actions_batch[0, 0], seed=1)
codes = th.zeros(N * n, V * V, dtype=th.uint8, device="cuda")
r = th.arange(N * n, device="cuda")
if os.environ.get("SAME_TILES"):
    codes[r, ((r % 16) * 14 + 3) % (V * V)] = 1          # the same 16 windows in every tile: what remains is the tile's position
else:
    codes[r, r % (V * V)] = (1 + r // (V * V)).to(th.uint8)
codes = codes.view(N, n, V, V)
dense = th.nn.functional.pad(codes.reshape(N, n, V * V), (0, abi.code_agent_stride(V) - V * V)).contiguous()
fp.encode(None, codes=dense, mask_alphabet=False)
out = fp.inputs[..., :32].clone().transpose(0, 1).reshape(N * n, 32)     # row = b * n + i
with th.no_grad():
    ref = ctx.mac.encode_obs(ctx.mac.expand_codes(codes)).reshape(N * n, 32)
d = (out - ref).abs().max(dim=1)[0].cpu().numpy()
bad = np.nonzero(d > 2e-6)[0]
print("rows off:", len(bad))
print("per 16-row tile: max diff")
for t0 in range(0, N * n, 16):
    print("  rows %4d..%4d (tile %d of its workgroup)  %.2e" % (t0, t0 + 15, (t0 // 16) % 4, d[t0:t0 + 16].max()))

# the same windows through the fused launch (inc head of t + encoder of t + 1)
g = th.Generator(device="cuda").manual_seed(1)
A = ctx.mac.args.n_actions
act = th.randint(0, A, (N, n), generator=g, device="cuda")
pos = th.rand(N, n, 2, generator=g, device="cuda") * 10
orient = th.zeros(N, n, 2, device="cuda")
reward = th.zeros(N, n, device="cuda"); clean = th.zeros(N, n, device="cuda"); den = th.rand(N, n, generator=g, device="cuda")
eps, step = th.full((), 0.3, device="cuda"), th.full((1,), 17, dtype=th.long, device="cuda")
fp.inputs_pair.zero_()
q = th.zeros(n, N, n, 3, device="cuda")
masks = th.tensor([0, 2, 1, 4], dtype=th.uint8, device="cuda")[dense.long()]    # the env's side buffer holds channel masks
fp.act_inc_encode(act, pos, orient, reward, clean, den, eps, step, masks, buf=0, q_out=q)
out2 = fp.inputs_pair[1][..., :32].clone().transpose(0, 1).reshape(N * n, 32)
d2 = (out2 - ref).abs().max(dim=1)[0].cpu().numpy()
print("fused launch: rows off %d" % int((d2 > 2e-6).sum()))
for t0 in range(0, N * n, 16):
    if d2[t0:t0 + 16].max() > 2e-6:
        print("  rows %4d..%4d (tile %d of its workgroup)  %.2e" % (t0, t0 + 15, (t0 // 16) % 4, d2[t0:t0 + 16].max()))
print("fused == standalone bit for bit:", bool(th.equal(out2, out)))
