"""Diagnostic: where does one wave spend its cycles?  Builds the kernels with -DSSD_STAMPS into
gpurun_out/libssd_hip_stamps.so, runs a few fused step+observe launches and prints the median cycles per phase.
(Stamps drain the wave's memory queue, so read SHARES, not totals.)  Usage: python tools/stamps.py [--n-env 4096]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-env", type=int, default=4096)
ap.add_argument("--mode", default="step_observe")
ap.add_argument("--dest", default="dense", choices=["dense", "storage"])
ap.add_argument("--fmt", default="f32", choices=["f32", "code"], help="observation format: f32 planes (format R) or u8 class codes (format C, the rollout's)")
ap.add_argument("--env-kind", default="cleanup")
ap.add_argument("--light", action="store_true", help="stamps wait for the LDS / scalar queue only, not for loads and stores in flight (-DSSD_STAMPS=2)")
a = ap.parse_args()
out = os.path.join(ROOT, "gpurun_out", "libssd_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(G.CSRC, s) for s in G.HIP_SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + G.HIPCC_FLAGS + ["-DSSD_STAMPS=%d" % (2 if a.light else 1), "-o", out] + srcs)
os.environ["SSD_HIP_LIB_PATH"] = out
import torch  # noqa: E402
from homophily_marl_amd import abi  # noqa: E402
from homophily_marl_amd.envs.native import NativeEnv  # noqa: E402

N, n = a.n_env, 5
env = NativeEnv("cleanup", device=0, map="default5", num_agents=n, n_env=N, view_size=7, episode_limit=100, rng_mode=abi.RNG_COUNTER, seed=1)
stamps = torch.zeros(N, 32, dtype=torch.int64, device="cuda")
env.lib.ssd_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
env.lib.ssd_debug_set_stamps(env.h, stamps.data_ptr())
avail = torch.tensor([0, 1, 2, 3, 4, 8], dtype=torch.int32, device="cuda")
env.reset()
out = None
FMT = abi.OBS_CODE if a.fmt == "code" else abi.OBS_F32
if a.dest == "storage":      # the rollout's destination: slot ep_step of an episode storage (every launch lands in fresh HBM lines)
    store = torch.empty(N, 101, n, 15, 15, dtype=torch.uint8, device="cuda") if a.fmt == "code" else torch.empty(N, 101, n, 3, 15, 15, device="cuda")
    out = env.storage_obs_buffers(store, FMT)
names = ["load", "moves", "consume+paint", "beams", "spawn", "scalars", "writeback", "obs pass0", "obs pass1", "obs pass2"]
acc = []
for t in range(40):
    acts = avail[torch.randint(0, 6, (N, n), device="cuda")].contiguous()
    if a.mode == "step_observe":
        env.step_observe(acts, fmt=FMT, out=out)
    else:
        env.step(acts)
    torch.cuda.synchronize()
    if t >= 20:
        acc.append(stamps.clone())
s = torch.stack(acc).double()            # [iters, N, 16]
d = s[..., 1:11] - s[..., 0:10]
tot = s[..., 10] - s[..., 0]
print("per-wave cycles (s_memtime ticks = shader cycles), median / p90 over waves and launches:")
for i, nm in enumerate(names):
    x = d[..., i].flatten()
    print("  %-14s %8.0f %8.0f" % (nm, x.median().item(), x.quantile(0.9).item()))
print("  %-14s %8.0f %8.0f" % ("total", tot.flatten().median().item(), tot.flatten().quantile(0.9).item()))
for nm, x in (("load: kernel arguments", s[..., 16] - s[..., 0]), ("load: requests issued", s[..., 17] - s[..., 16]), ("load: counters landed, windows requested, grid in LDS", s[..., 18] - s[..., 17]),
              ("load: rest", s[..., 1] - s[..., 18]), ("scalars: table look-up", s[..., 19] - s[..., 5]), ("scalars: output pointers", s[..., 20] - s[..., 19]), ("scalars: stores", s[..., 6] - s[..., 20])):
    x = x.flatten()
    print("  %-54s %8.0f %8.0f" % (nm, x.median().item(), x.quantile(0.9).item()))
if a.env_kind == "cleanup":
    for nm, x in (("spawn: apples", s[..., 14] - s[..., 4]), ("spawn: waste keys + J", s[..., 15] - s[..., 14]), ("spawn: selection", s[..., 5] - s[..., 15])):
        x = x.flatten()
        print("  %-22s %8.0f %8.0f" % (nm, x.median().item(), x.quantile(0.9).item()))

# timeline of ONE launch on the chip-wide 100 MHz clock (10 ns ticks): when do waves start (dispatch ramp) and end (tail)?
one = acc[-1].cpu()
t0, t1 = one[:, 12].double(), one[:, 13].double()
base = t0.min()
q = torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.double)
print("one launch, microseconds after the first wave's start (min / p10 / median / p90 / p99 / max):")
print("  wave start   ", " ".join("%6.2f" % (x / 100) for x in torch.quantile(t0 - base, q).tolist()))
print("  wave end     ", " ".join("%6.2f" % (x / 100) for x in torch.quantile(t1 - base, q).tolist()))
print("  wave lifetime", " ".join("%6.2f" % (x / 100) for x in torch.quantile(t1 - t0, q).tolist()))
xcc = (one[:, 11] & 15)
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("  XCC %d: %4d waves, start %6.2f..%6.2f, end %6.2f..%6.2f" % (x, int(m.sum()), (t0[m].min() - base) / 100, (t0[m].max() - base) / 100,
                                                                     (t1[m].min() - base) / 100, (t1[m].max() - base) / 100))
late = torch.argsort(t1)[-8:]
print("  last 8 waves: env", late.tolist(), "start", [round((t0[i].item() - base.item()) / 100, 2) for i in late], "life", [round((t1[i] - t0[i]).item() / 100, 2) for i in late])
# which phases are long in the slowest 1% of waves?
life = (one[:, 10] - one[:, 0]).double()
slow = life >= torch.quantile(life, torch.tensor(0.99, dtype=torch.double))
dd = (one[:, 1:11] - one[:, 0:10]).double()
print("  phase medians, all waves vs the slowest 1%:")
for i, nm in enumerate(names):
    print("    %-14s %8.0f %8.0f" % (nm, dd[:, i].median().item(), dd[slow, i].median().item()))
