"""Diagnostic: k_env<STEP_OBS> launch time by observation destination -- dense buffer (what `bench.py --workload env` times),
episode storage [n_env, T+1, ...] (env-major, what the rollout writes), time-major storage, each with and without the class-code
side output.    python3 tools/envprof.py [--config cleanup5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from bench import CONFIGS  # noqa: E402
from homophily_marl_amd import abi  # noqa: E402
from homophily_marl_amd.envs.native import NativeEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cleanup5", choices=sorted(CONFIGS))
ap.add_argument("--fmt", default="f32")
args = ap.parse_args()
c = CONFIGS[args.config]
N, n, T, V = c["n_env"], c["n_agents"], 100, 2 * c["view_size"] + 1
dev = th.device("cuda", 0)
nat = NativeEnv(c["env"], device=0, map=c["map"], num_agents=n, n_env=N, view_size=c["view_size"], episode_limit=T,
                rng_mode=abi.RNG_COUNTER, seed=1)
avail = th.tensor([0, 1, 2, 3, 4, 8] if c["env"] == "cleanup" else [0, 1, 2, 3, 4], dtype=th.int32, device=dev)
acts = [avail[th.randint(0, avail.numel(), (N, n), device=dev)].contiguous() for _ in range(8)]
fmt = {"f32": abi.OBS_F32, "code": abi.OBS_CODE, "u8": abi.OBS_U8}[args.fmt]
dt = {"f32": th.float32, "code": th.uint8, "u8": th.uint8}[args.fmt]
inner = (n, V, V) if fmt == abi.OBS_CODE else (n, 3, V, V)
store_em = th.empty((N, T + 1) + inner, dtype=dt, device=dev)                  # env-major: the EpisodeBatch layout
store_tm = th.empty((T + 1, N) + inner, dtype=dt, device=dev).transpose(0, 1)  # time-major memory, same logical shape
variants = [("dense", lambda wc: nat.obs_buffers(fmt, want_code=wc)),
            ("storage env-major", lambda wc: nat.storage_obs_buffers(store_em, fmt, want_code=wc)),
            ("storage time-major", lambda wc: nat.storage_obs_buffers(store_tm, fmt, want_code=wc))]
for name, mk in variants:
    for wc in ([False, True] if fmt != abi.OBS_CODE else [False]):
        try:
            out = mk(wc)
        except AssertionError as e:
            print("%-22s code=%d: refused (%s)" % (name, wc, e))
            continue
        per = []
        for rep in range(4):
            nat.reset()
            for i in range(10):
                nat.step_observe(acts[i % 8], None, fmt, out=out)
            a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            a.record()
            for i in range(80):
                nat.step_observe(acts[i % 8], None, fmt, out=out)
            b.record()
            th.cuda.synchronize()
            per.append(1e3 * a.elapsed_time(b) / 80)
        assert nat.poll_error() == 0
        print("%-22s code=%d: %s us / launch" % (name, wc, " ".join("%.2f" % x for x in per)), flush=True)
