"""Kernel micro-benchmark: per-launch time of ssd_step / ssd_observe / ssd_step_observe (all formats) on one GPU.
Usage: python tools/kbench.py [--n-env 4096] [--env cleanup --map default5 --agents 5 --view 7] [--iters 300]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from homophily_marl_amd import abi  # noqa: E402
from homophily_marl_amd.envs.native import NativeEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-env", type=int, default=4096)
ap.add_argument("--env", default="cleanup"); ap.add_argument("--map", default="default5")
ap.add_argument("--agents", type=int, default=5); ap.add_argument("--view", type=int, default=7)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--only", default="")
a = ap.parse_args()
N, n = a.n_env, a.agents
env = NativeEnv(a.env, device=0, map=a.map, num_agents=n, n_env=N, view_size=a.view, episode_limit=100, rng_mode=abi.RNG_COUNTER, seed=1)
dev = env.device
avail = torch.tensor([0, 1, 2, 3, 4, 8] if a.env == "cleanup" else [0, 1, 2, 3, 4], dtype=torch.int32, device=dev)
acts = [avail[torch.randint(0, len(avail), (N, n), device=dev)].contiguous() for _ in range(16)]


def timeit(name, fn):
    env.reset()
    for t in range(20):
        fn(t)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for t in range(a.iters):
        if t % 100 == 0:
            env.reset()
        fn(t)
    e.record(); torch.cuda.synchronize()
    print("%-28s %8.2f us/launch (back-to-back, incl. gaps)" % (name, 1e3 * s.elapsed_time(e) / a.iters), flush=True)


cases = {
    "step": lambda t: env.step(acts[t % 16]),
    "observe_f32": lambda t: env.observe(abi.OBS_F32),
    "observe_bf16": lambda t: env.observe(abi.OBS_BF16),
    "observe_u8": lambda t: env.observe(abi.OBS_U8),
    "observe_code": lambda t: env.observe(abi.OBS_CODE),
    "step_observe_f32": lambda t: env.step_observe(acts[t % 16], fmt=abi.OBS_F32),
    "step_observe_bf16": lambda t: env.step_observe(acts[t % 16], fmt=abi.OBS_BF16),
    "step_observe_code": lambda t: env.step_observe(acts[t % 16], fmt=abi.OBS_CODE),
}
for k, f in cases.items():
    if not a.only or k in a.only.split(","):
        timeit(k, f)
assert env.poll_error() == 0
