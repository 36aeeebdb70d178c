"""Diagnostic: run rollout episodes only (no training) so a rocprofv3 --kernel-trace --stats of this script shows the
per-timestep kernel mix of the hip_graph runner.  Usage: python tools/rollout_prof.py [--episodes 6] [--runner hip_graph]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as th
from homophily_marl_amd.run import load_config, setup
ap = argparse.ArgumentParser()
ap.add_argument("--episodes", type=int, default=6); ap.add_argument("--runner", default="hip_graph"); ap.add_argument("--n-env", type=int, default=4096)
a = ap.parse_args()
cfg = load_config("cleanup", overrides=dict(runner=a.runner, batch_size_run=a.n_env, batch_size=16, buffer_size=16, buffer_cpu_only=False,
                                             store_state=False, env_args=dict(num_agents=5, map="default5", episode_limit=100, seed=1),
                                             use_cuda=True, save_model=False, runner_stats=False))
th.manual_seed(0)
ctx = setup(cfg)
for ep in range(a.episodes):
    th.cuda.synchronize(); t = time.perf_counter()
    ctx.runner.run(False)
    th.cuda.synchronize(); print("episode %d: %.1f ms" % (ep, 1e3 * (time.perf_counter() - t)), flush=True)
