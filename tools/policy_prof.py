"""Time the rollout-time policy pieces on the GPU: fused heads vs the per-layer composition (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from homophily_marl_amd.fast_policy import FastPolicy
from homophily_marl_amd.run import load_config, setup

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
avail = env.avail_actions_batch[0, 0]
for fused in (False, True):
    fp = FastPolicy(mac, N, avail, seed=7, fused=fused)
    def env_head():
        return fp.act_env(obs, prev_a, prev_r, prev_i, pos, eps, step)
    def inc_head():
        return fp.act_inc(fp.actions, pos, orient, prev_r, prev_r, prev_r, eps, step)
    for name, fn in (("act_env", env_head), ("act_inc", inc_head)):
        for _ in range(5):
            fn()
        th.cuda.synchronize()
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record(); th.cuda.synchronize()
        print("fused=%d %s: %.1f us/call (eager, incl. launch gaps)" % (fused, name, 1e3 * a.elapsed_time(b) / 50), flush=True)
