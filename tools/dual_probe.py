"""Diagnostic: do two independent half-batch rollouts on two streams overlap on the chip?  Two complete runners of N / 2 envs each
(own env handle, own FastPolicy, own graphs; env_id_base 0 / N / 2) are replayed from one host thread on two streams; compared with one
runner of N envs.  Rollout only (no learner)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from homophily_marl_amd.run import load_config, setup  # noqa: E402

N, n, T = int(os.environ.get("N", 4096)), 5, 100
PARTS = int(os.environ.get("PARTS", 2))


def make(n_env, base):
    cfg = load_config("cleanup", overrides=dict(
        runner="hip_graph", steps_per_graph=10, batch_size_run=n_env, batch_size=16, buffer_size=n_env, obs_storage="code",
        buffer_cpu_only=False, store_state=False, env_args=dict(num_agents=n, map="default5", episode_limit=T, seed=1),
        use_cuda=True, save_model=False, env_id_base=base, runner_stats=False, learner_log_interval=10 ** 12))
    th.manual_seed(0)
    return setup(cfg)


def rollout(ctxs, streams):
    for c, s in zip(ctxs, streams):
        with th.cuda.stream(s):
            c.runner.begin_episode(False)
    for t in range(T):
        for c, s in zip(ctxs, streams):
            with th.cuda.stream(s):
                c.runner.step_once()
    for c, s in zip(ctxs, streams):
        with th.cuda.stream(s):
            c.runner.finish_episode()


def timed(ctxs, streams, reps=15):
    for _ in range(5):
        rollout(ctxs, streams)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rollout(ctxs, streams)
    th.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


one = make(N, 0)
print("1 runner x %d envs: %.3f ms per rollout" % (N, timed([one], [th.cuda.current_stream()])), flush=True)
del one
parts = [make(N // PARTS, k * (N // PARTS)) for k in range(PARTS)]
streams = [th.cuda.Stream() for _ in range(PARTS)]
print("%d runners x %d envs on %d streams: %.3f ms per rollout of all %d envs" % (PARTS, N // PARTS, PARTS, timed(parts, streams), N), flush=True)
print("the same %d runners one after the other on one stream: %.3f ms" % (PARTS, timed(parts, [th.cuda.current_stream()] * PARTS)), flush=True)
