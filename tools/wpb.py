"""Diagnostic: fused-kernel time for different numbers of envs (waves) per workgroup."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as G
srcs = [os.path.join(G.CSRC, s) for s in G.HIP_SOURCES]
for w in (1, 2, 4, 8, 16):
    out = os.path.join(ROOT, "gpurun_out", "libssd_hip_wpb%d.so" % w)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + G.HIPCC_FLAGS + ["-DSSD_WAVES_PER_BLOCK=%d" % w, "-o", out] + srcs, stderr=subprocess.DEVNULL)
    print("--- waves per block =", w, flush=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--only", "step,step_observe_f32"], env=dict(os.environ, SSD_HIP_LIB_PATH=out))
