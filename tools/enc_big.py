"""Diagnostic: the fused encoder at large row counts against torch in float64 (exact reference) and float32.
python tools/enc_big.py N [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch as th
from test_policy_mfma import _ctx, _random_codes
from homophily_marl_amd import abi
from homophily_marl_amd.fast_policy import FastPolicy

N = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
V = 15
th.manual_seed(2)
ctx = _ctx("cleanup", n, N, view=7)
mac = ctx.mac
fp = FastPolicy(mac, N, ctx.runner.env.avail_actions_batch[0, 0], seed=1)
g = th.Generator(device="cuda").manual_seed(5)
codes = _random_codes(N, n, V, g)
obs = mac.expand_codes(codes)
enc = mac.agent.conv_to_fc
with th.no_grad():
    ref32 = mac.encode_obs(obs).reshape(N, n, 32).transpose(0, 1)
    e64 = __import__("copy").deepcopy(enc).double()
    ref64 = e64(obs.reshape(N * n, 3, V, V).double()).reshape(N, n, 32).transpose(0, 1)
    # torch f32 in chunks of 1024 rows (does the library pick another algorithm at large batch?)
    chunks = th.cat([mac.encode_obs(obs[i:i + 128]) for i in range(0, N, 128)]).reshape(N, n, 32).transpose(0, 1)
dense = th.nn.functional.pad(codes.reshape(N, n, V * V), (0, abi.code_agent_stride(V) - V * V)).contiguous()
fp.encode(None, codes=dense, mask_alphabet=False)
out = fp.inputs[..., :32].clone()
print("rows", N * n, "SSD_ENC_BT", os.environ.get("SSD_ENC_BT"))
print("kernel  vs f64: %.3e" % (out.double() - ref64).abs().max().item())
print("torch32 vs f64: %.3e" % (ref32.double() - ref64).abs().max().item())
print("torch32 (128-env chunks) vs f64: %.3e" % (chunks.double() - ref64).abs().max().item())
d = (out.double() - ref64).abs().amax(dim=2)        # [n, N]
bad = (d > 2e-6)
print("kernel rows off: %d of %d" % (int(bad.sum()), bad.numel()))
if bad.any():
    idx = th.nonzero(bad)
    print("first bad (agent, env):", idx[:10].tolist(), "last:", idx[-5:].tolist())
    rows = (idx[:, 1] * n + idx[:, 0])          # row = b * n + i
    print("bad row range: %d .. %d; rows mod 80 histogram:" % (int(rows.min()), int(rows.max())), th.bincount(rows % 80, minlength=80).tolist())
