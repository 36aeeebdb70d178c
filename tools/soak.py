"""Diagnostic: a short real training run (hip_graph runner, in-place replay, graph-captured train step); prints the learner's
losses and the mean episode return every few iterations and checks that everything stays finite."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as th
from homophily_marl_amd.run import load_config, setup, train_iteration

N, iters = int(os.environ.get("N_ENV", 1024)), int(os.environ.get("ITERS", 60))
cfg = load_config("cleanup", overrides=dict(runner="hip_graph", train_graph=1, batch_size_run=N, batch_size=16, buffer_size=4 * N,
                                             buffer_cpu_only=False, store_state=False,
                                             env_args=dict(num_agents=5, map="default5", episode_limit=100, seed=1),
                                             use_cuda=True, save_model=False, runner_stats=False, learner_log_interval=10 ** 12))
th.manual_seed(0)
ctx = setup(cfg)
ep, t0 = 0, time.time()
for it in range(iters):
    ep = train_iteration(ctx, ep)
    if it % 10 == 9 or it == iters - 1:
        th.cuda.synchronize()
        logs = {k: float(v) for k, v in ctx.learner._static_logs.items()} if getattr(ctx.learner, "_static_logs", None) else {}
        ret = float(ctx.runner.ep_return.sum(-1).mean())
        p = th.cat([q.detach().flatten() for q in ctx.mac.parameters()])
        print("iter %3d  t_env %8d  mean collective return %7.2f  loss_env %.5f loss_inc %.5f loss_sim %.5f  |theta| %.3f  finite %s  %.1fs" % (
            it + 1, ctx.runner.t_env, ret, logs.get("loss_value_env", float("nan")), logs.get("loss_value_inc", float("nan")),
            logs.get("loss_sim", float("nan")), float(p.norm()), bool(th.isfinite(p).all()), time.time() - t0), flush=True)
assert ctx.runner.env.native.poll_error() == 0
