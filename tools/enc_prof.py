"""Diagnostic: torch / MIOpen encoder (conv_to_fc) forward + backward time at the learner's batch (16 x 101 x 5 rows)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as th
import torch.nn as nn
R = int(os.environ.get("ROWS", 8080))
enc = nn.Sequential(nn.Conv2d(3, 6, 3, 1), nn.LeakyReLU(), nn.Flatten(), nn.Linear(1014, 32), nn.LeakyReLU()).cuda()
x = th.rand(R, 3, 15, 15, device="cuda")
def ev():
    return th.cuda.Event(enable_timing=True)
for mode in ("fwd_nograd", "fwd", "fwd+bwd"):
    for it in range(8):
        if it == 3:
            th.cuda.synchronize(); a, b = ev(), ev(); a.record()
        if mode == "fwd_nograd":
            with th.no_grad():
                y = enc(x)
        else:
            y = enc(x)
            if mode == "fwd+bwd":
                y.sum().backward()
    b.record(); th.cuda.synchronize()
    print("%s: %.3f ms/call" % (mode, a.elapsed_time(b) / 5), flush=True)
