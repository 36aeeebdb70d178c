"""Diagnostic: per-iteration wall time of run.train_iteration (synchronised after each) to spot periodic host-side costs.
python tools/iter_times.py [tspr] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from homophily_marl_amd.run import load_config, setup, train_iteration
tspr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = 4096
cfg = load_config("cleanup", overrides=dict(runner="hip_graph", train_graph=1, steps_per_graph=10, batch_size_run=N, batch_size=16, buffer_size=2 * N,
                                             obs_storage="code", buffer_cpu_only=False, store_state=False, train_steps_per_rollout=tspr,
                                             env_args=dict(num_agents=5, map="default5", episode_limit=100, seed=1), use_cuda=True, save_model=False,
                                             strict_device_ops=True))
th.manual_seed(0)
ctx = setup(cfg)
ep = 0
cold = []
for _ in range(int(os.environ.get("COLD", 6))):
    t0 = time.perf_counter()
    ep = train_iteration(ctx, ep)
    th.cuda.synchronize()
    cold.append(1e3 * (time.perf_counter() - t0))
print("cold-start iterations ms:", " ".join("%.1f" % t for t in cold))
ts = []
import cProfile, pstats
for i in range(iters):
    t0 = time.perf_counter()
    ep = train_iteration(ctx, ep)
    th.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print("per-iteration ms:", " ".join("%.1f" % t for t in ts))
print("mean %.2f  median %.2f  max %.2f" % (sum(ts) / len(ts), sorted(ts)[len(ts) // 2], max(ts)))
# unsynchronised loop (as the bench times it)
th.cuda.synchronize(); t0 = time.perf_counter()
for i in range(iters):
    ep = train_iteration(ctx, ep)
th.cuda.synchronize()
print("pipelined: %.2f ms / iteration" % (1e3 * (time.perf_counter() - t0) / iters))
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    ep = train_iteration(ctx, ep)
th.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
