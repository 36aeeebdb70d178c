"""Diagnostic: phase stamps (s_memtime) of the fused policy kernels.  Builds -DSSD_STAMPS into gpurun_out/libssd_hip_stamps.so."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libssd_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(G.CSRC, s) for s in G.HIP_SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + G.HIPCC_FLAGS + ["-DSSD_STAMPS", "-o", out] + srcs)
os.environ["SSD_HIP_LIB_PATH"] = out
import torch as th  # noqa: E402
from homophily_marl_amd import abi  # noqa: E402
from homophily_marl_amd.fast_policy import FastPolicy  # noqa: E402
from homophily_marl_amd.run import load_config, setup  # noqa: E402

N, n = int(os.environ.get("N_ENV", 4096)), 5
cfg = load_config("cleanup", overrides=dict(runner="hip_vec", batch_size_run=N, batch_size=8, buffer_size=8, buffer_cpu_only=False,
                                             store_state=False, env_args=dict(num_agents=n, map="default5", episode_limit=20, seed=3),
                                             use_cuda=True, save_model=False, runner_stats=False))
ctx = setup(cfg)
mac, env = ctx.mac, ctx.runner.env
env.reset_batch()
o = env.observe_batch()
obs, pos, orient = o["obs"], o["pos"], o["orient"]
g = th.Generator(device="cuda").manual_seed(0)
prev_a = th.randint(-1, 9, (N, n), generator=g, device="cuda")
prev_r = th.zeros(N, n, device="cuda"); prev_i = th.zeros(N, n, n, dtype=th.long, device="cuda")
eps, step = th.zeros((), device="cuda"), th.zeros(1, dtype=th.long, device="cuda")
fp = FastPolicy(mac, N, env.avail_actions_batch[0, 0], seed=7, fused=True)
lib = abi.load_library()
lib.ssd_debug_set_policy_stamps.argtypes = [C.c_void_p]
stamps = th.zeros(8192, 16, dtype=th.int64, device="cuda")
store = th.zeros(N, 2, n, 3, 15, 15, device="cuda"); tdev = th.zeros(1, dtype=th.long, device="cuda")


def show(name, labels, nwaves, sums=()):
    s = stamps[:nwaves].double()
    live = s[:, labels[-1][0]] > 0
    s = s[live]
    t0 = s[:, 0].min()
    print("%s: %d waves stamped; kernel span %.0f cycles" % (name, s.shape[0], (s[:, labels[-1][0]].max() - t0).item()))
    prev = 0
    for idx, nm in labels[1:]:
        ok = s[:, idx] > 0
        d = (s[ok, idx] - s[ok, prev])
        print("  %-22s median %8.0f  p90 %8.0f   (waves %d)" % (nm, d.median().item(), d.quantile(0.9).item(), int(ok.sum())))
        prev = idx
    for idx, nm in sums:
        print("  %-22s median %8.0f  p90 %8.0f" % (nm, s[:, idx].median().item(), s[:, idx].quantile(0.9).item()))
    print("  start skew (last wave start - first): %.0f" % (s[:, 0].max() - t0).item())


for rep in range(3):
    stamps.zero_(); lib.ssd_debug_set_policy_stamps(stamps.data_ptr())
    lib_args = None
    # encoder alone
    abi.check(lib, lib.ssd_policy_encode(obs.data_ptr(), N * n, 15, fp.p["cw"].data_ptr(), fp.p["cb"].data_ptr(), fp.p["lwp"].data_ptr(),
                                         fp.p["lb"].data_ptr(), fp.inputs.data_ptr(), 64, n, 1, 0, 0, None, None, None,
                                         th.cuda.current_stream().cuda_stream))
    th.cuda.synchronize()
    if rep == 2:
        show("k_encode", [(0, "start"), (1, "stage+sync"), (2, "conv+linear groups"), (3, "reduce+out")],
             N * n // 16 * 4)
    stamps.zero_()
    lib.ssd_debug_set_policy_stamps(None)
    fp.act_env(obs, prev_a, prev_r, prev_i, pos, eps, step)            # encoder unstamped, then the head stamped
    th.cuda.synchronize()
    stamps.zero_(); lib.ssd_debug_set_policy_stamps(stamps.data_ptr())
    fp.act_inc(fp.actions, pos, orient, prev_r, prev_r, prev_r, eps, step)
    th.cuda.synchronize()
    if rep == 2:
        show("k_head<inc>", [(0, "start"), (1, "stage weights+sync"), (2, "inputs"), (3, "fc1"), (4, "gru Wi"), (5, "gru Wh"), (6, "gates+store"),
                             (7, "fc2+pick"), (8, "other tiles+end")], 255 * 8)
