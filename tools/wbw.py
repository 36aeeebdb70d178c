"""Diagnostic: raw streaming-write time for an obs-sized buffer, and kbench against a -DSSD_NOSTORE build."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import torch
x = torch.empty(4096 * 5 * 675, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn in (("fill_(55MB)", lambda: x.fill_(1.0)), ("copy_(55MB->55MB)", lambda: y.copy_(x))):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): fn()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 200
    print("%-20s %7.2f us  -> %.2f TB/s written" % (name, us, x.numel() * 4 / us / 1e6))
out = os.path.join(ROOT, "gpurun_out", "libssd_hip_nostore.so")
srcs = [os.path.join(G.CSRC, s) for s in G.HIP_SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + G.HIPCC_FLAGS + ["-DSSD_NOSTORE", "-o", out] + srcs, stderr=subprocess.DEVNULL)
env = dict(os.environ, SSD_HIP_LIB_PATH=out)
print("--- kbench with obs stores suppressed (compute only) ---", flush=True)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--only", "observe_f32,observe_u8,step_observe_f32"], env=env)
