"""Diagnostic: timeline of the rollout controller kernels from in-kernel s_memtime stamps.  Builds the kernels with -DSSD_STAMPS
into gpurun_out/libssd_hip_stamps.so (the product library has no stamp code), runs each kernel a few times on live data and
prints, per kernel, the distribution over waves of the phase boundaries relative to the first wave's start (cycles of the
100 MHz-independent shader clock; s_memtime).  Usage: python tools/pstamps.py [--config cleanup5]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cleanup5")
a = ap.parse_args()
out = os.path.join(ROOT, "gpurun_out", "libssd_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(G.CSRC, s) for s in G.HIP_SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + G.HIPCC_FLAGS + ["-DSSD_STAMPS", "-o", out] + srcs)
os.environ["SSD_HIP_LIB_PATH"] = out
import torch as th  # noqa: E402
from bench import CONFIGS  # noqa: E402
from homophily_marl_amd import abi  # noqa: E402
from homophily_marl_amd.run import load_config, setup  # noqa: E402

c = CONFIGS[a.config]
N, n, T = c["n_env"], c["n_agents"], 100
cfg = load_config(c["env"], overrides=dict(
    runner="hip_graph", rollout_graph=False, batch_size_run=N, batch_size=16, buffer_size=N, buffer_cpu_only=False, store_state=False,
    pipeline_encode=False,      # the phase stamps are defined per standalone kernel (k_encode, k_head<env>, k_head<inc>)
    env_args=dict(num_agents=n, map=c["map"], episode_limit=T, view_size=c["view_size"], seed=1),
    use_cuda=True, save_model=False, runner_stats=False, learner_log_interval=10 ** 12))
th.manual_seed(0)
ctx = setup(cfg)
r = ctx.runner
lib = abi.load_library()
lib.ssd_debug_set_policy_stamps.argtypes = [C.c_void_p]
r.begin_episode(False)
for _ in range(12):
    r.step_once()
th.cuda.synchronize()
NW = 4096 * 16
stamps = th.zeros(NW, 16, dtype=th.int64, device="cuda")
names = {"encode": ["start", "staged", "units done", "end"],
         "head_env": ["start", "staged", "inputs", "fc1", "gru mfma", "gates", "tile done", "end"],
         "head_inc": ["start", "staged", "inputs", "fc1", "gru mfma", "gates", "tile done", "end"]}
for name, key, fn in r.timestep_launches():
    if key == "env":
        continue
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    lib.ssd_debug_set_policy_stamps(stamps.data_ptr())
    stamps.zero_()
    fn()
    th.cuda.synchronize()
    lib.ssd_debug_set_policy_stamps(None)
    s = stamps.cpu()
    s = s[s[:, 0] > 0]
    # s_memtime counters differ between XCDs: report per-WAVE phase durations (and every wave's lifetime), not absolute times
    print("%s: %d waves" % (name, s.shape[0]))
    labels = names[key]
    last = len(labels) - 1
    life = (s[:, last] - s[:, 0]).double()
    print("   %-22s waves %5d  min %7d  median %7d  p90 %7d  max %7d" % ("wave lifetime", life.numel(), *[int(x) for x in th.quantile(life, th.tensor([0.0, 0.5, 0.9, 1.0], dtype=th.double))]))
    for i in range(1, len(labels)):
        ok = (s[:, i] > 0) & (s[:, i - 1] > 0)
        if i == last:            # "end" follows the last per-tile stamp only for waves that had a tile
            prev = th.where(s[:, last - 1] > 0, s[:, last - 1], s[:, 1])
            d = (s[:, i] - prev).double()
        else:
            d = (s[ok, i] - s[ok, i - 1]).double()
        if d.numel():
            qs = th.quantile(d, th.tensor([0.0, 0.5, 0.9, 1.0], dtype=th.double))
            print("   %-22s waves %5d  min %7d  median %7d  p90 %7d  max %7d" % (labels[i - 1] + " -> " + labels[i], d.numel(), *[int(x) for x in qs]))
    # timeline of the launch on the chip-wide 100 MHz clock (10 ns ticks)
    t0, t1 = s[:, 14].double(), s[:, 15].double()
    base = t0.min()
    q = th.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=th.double)
    print("   us after the first wave's start (min / p10 / median / p90 / max): start %s | end %s | kernel span %.2f us" % (
        " ".join("%.2f" % (x / 100) for x in th.quantile(t0 - base, q).tolist()),
        " ".join("%.2f" % (x / 100) for x in th.quantile(t1 - base, q).tolist()), (t1.max() - base).item() / 100))
    if key in ("head_env", "head_inc"):      # the prologue: scalars + issue of the tile's loads | their landing + input assembly | first chunks
        pro = ((0, 8, "start -> kernargs"), (8, 9, "1st DMA batch issued"), (9, 10, "tile loads issued"), (10, 11, "scalars + avail + ALL landed"), (11, 12, "-> prepare"))
        for a_, b_, nm in (pro if key == "head_env" else ()) + ((0, 12, "start -> loads issued"), (12, 13, "loads landed + prepare"), (13, 1, "2nd DMA batch + wait + barrier")):
            ok = (s[:, a_] > 0) & (s[:, b_] > 0)
            d = (s[ok, b_] - s[ok, a_]).double()
            if d.numel():
                print("   %-32s waves %5d  median %7d  p90 %7d" % (nm, d.numel(), int(d.median()), int(th.quantile(d, th.tensor(0.9, dtype=th.double)))))
    if key == "head_inc":
        for a_, b_, nm in ((4, 5, "gru mfma -> gates"), (5, 8, "gates -> fc2 + scratch"), (8, 9, "cold pointer loads"), (9, 10, "items k = 0"), (10, 11, "items k = 1"), (11, 6, "k = 2 (empty) -> tile done")):
            ok = (s[:, a_] > 0) & (s[:, b_] > 0)
            d = (s[ok, b_] - s[ok, a_]).double()
            if d.numel():
                print("   %-28s waves %5d  median %7d  p90 %7d" % (nm, d.numel(), int(d.median()), int(th.quantile(d, th.tensor(0.9, dtype=th.double)))))
