"""End-to-end workload of bench.py: the full homophily training loop on vectorised envs.

One "step" = one transition of all N envs of this rank inside the real loop: observation storage, env-head and
incentive-head action selection (Q-net forward), the fused ssd_step_observe launch, and -- at every episode end --
slot-T bootstrapping, replay insertion, sampling and ONE learner.train (run.py:184-210 cadence, batch_size 16).
Nothing is skipped inside the timed region.
"""
import time

import torch as th
import torch.distributed as dist

from . import abi
from .run import load_config, setup


def run_e2e(args, rank, world, local_rank):
    N, n, T = args.n_env, 5, 100
    # replay capacity: at least the reference's 5000 episodes (config/default.yaml), rounded up to a multiple of the env batch so
    # that the runner can write its episodes straight into the buffer's slots (ReplayBuffer.reserve: insertion moves no data)
    buffer_size = -(-5000 // N) * N
    # timesteps per rollout-graph replay: only if the timed region and the warm-up are whole numbers of replays, so that
    # EXACTLY `steps` timesteps of work are inside the timed region
    spg = int(getattr(args, "steps_per_graph", 10))
    if spg < 1 or args.steps % spg or args.warmup % spg or T % spg:
        spg = 1
    cfg = load_config("cleanup", overrides=dict(
        runner=args.runner, train_graph=args.train_graph, steps_per_graph=spg, batch_size_run=N, batch_size=16, buffer_size=buffer_size, obs_storage=getattr(args, "obs_storage", "f32"), buffer_cpu_only=False, store_state=False,
        env_args=dict(num_agents=n, map="default5", episode_limit=T, view_size=7, seed=1), use_cuda=True, save_model=False,
        device_index=local_rank, env_id_base=rank * N, runner_stats=False, learner_log_interval=10 ** 12))
    th.manual_seed(0)                     # fixed-seed random-init weights, identical on every rank
    ctx = setup(cfg)
    runner, learner, buf = ctx.runner, ctx.learner, ctx.buffer
    a = ctx.args
    state = dict(episode=0, in_episode=False)

    def one_step():
        if not state["in_episode"]:
            runner.begin_episode(False)
            state["in_episode"] = True
        if runner.step_once():
            batch = runner.finish_episode()
            buf.insert_episode_batch(batch)
            if buf.can_sample(a.batch_size):
                sample = buf.sample(a.batch_size)
                sample = sample[:, :T + 1]
                learner.train(sample, runner.t_env, state["episode"])
            state["episode"] += a.batch_size_run
            state["in_episode"] = False

    # Setup (not warm-up): build the hipGraphs.  The rollout graph is captured at the start of the 2nd episode and the two
    # train-step graphs at the 3rd learner.train call, so at least 4 full iterations (every replay slab is visited) are run before the W warm-up steps; this is the
    # analogue of compiling the step and is excluded from both the warm-up count and the timed region.
    for _ in range(max(4, buffer_size // N + 2) * T):       # every replay slab gets its graph before the warm-up starts
        one_step()
    for _ in range(args.warmup):
        one_step()
    th.cuda.synchronize()
    if world > 1:
        dist.barrier()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    th.cuda.synchronize()
    if world > 1:
        dist.barrier()
    th.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert runner.env.native.poll_error() == 0
    # Steady-state breakdown of one more rollout + train iteration (synchronised between phases; NOT part of `elapsed`)
    def timed(fn):
        th.cuda.synchronize(); t = time.perf_counter(); r = fn(); th.cuda.synchronize()
        return r, 1e3 * (time.perf_counter() - t)
    while state["in_episode"]:
        one_step()
    bd = {}
    _, bd["begin_episode_ms"] = timed(lambda: runner.begin_episode(False))
    _, bd["rollout_100_steps_ms"] = timed(lambda: [runner.step_once() for _ in range(T)])
    batch, bd["finish_episode_ms"] = timed(runner.finish_episode)
    _, bd["replay_insert_ms"] = timed(lambda: buf.insert_episode_batch(batch))
    sample, bd["sample_ms"] = timed(lambda: buf.sample(a.batch_size)[:, :T + 1])
    _, bd["learner_train_ms"] = timed(lambda: learner.train(sample, runner.t_env, state["episode"]))
    bd = {k: round(v, 3) for k, v in bd.items()}
    # Kernel timing for the roofline: inside a hipGraph replay there is no host call to bracket, so the SAME kernel on the
    # SAME live env object is timed with HIP-event pairs (torch's current stream = the launch stream) right after the
    # timed region, over three whole episodes with actions drawn like the policy's epsilon-random ones.
    env = runner.env
    avail = th.nonzero(env.avail_actions_batch[0, 0]).squeeze(-1).to(th.int32)
    acts = [avail[th.randint(0, avail.numel(), (N, n), device=env.device)].contiguous() for _ in range(8)]
    code = getattr(args, "obs_storage", "f32") == "code"
    kfmt = abi.OBS_CODE if code else abi.OBS_F32   # the format the loop's env launches emit
    env.reset_batch()
    for i in range(20):                           # un-timed: the eager launch path has been idle during the graph replays
        env.step_batch(acts[i % 8], observe=True, fmt=kfmt)
    ev, run_len = [], []
    for rep in range(3):                          # three whole episodes, like the env workload: reset, then T back-to-back launches
        env.reset_batch()
        s, e = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        s.record()
        for i in range(T):                        # back-to-back launches of the dominant kernel, bracketed by two events
            env.step_batch(acts[i % 8], observe=True, fmt=kfmt)
        e.record()
        ev.append((s, e)); run_len.append(T)
    th.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) / k for (s, e), k in zip(ev, run_len))
    from bench import algorithmic_bytes_per_env_step
    return dict(elapsed=elapsed, kern_avg_us=1e3 * sum(ms) / len(ms), kern_med_us=1e3 * ms[len(ms) // 2],
                bytes_per_launch=(algorithmic_bytes_per_env_step(25, 18, n, 15, 1, 1) if code else algorithmic_bytes_per_env_step(25, 18, n, 15)) * N,
                dtype="fp32",
                workload="cleanup_default5_rollout_plus_homophily_train",
                extra=dict(obs_format="u8 class codes [n_env,n,15,15] (format C)" if code else "f32[n_env,n,3,15,15]", kernel="ssd::k_env<MODE_STEP_OBS>", qnet_dtype="fp32",
                           train="1 learner.train(batch_size 16 x T 101) per 100-step rollout, double-Q + sim loss, 2x Adam",
                           buffer="device-resident ReplayBuffer, %d episodes, %s" % (buf.buffer_size, "written in place by the runner" if getattr(runner, "_replay", None) is not None else "copy insertion"),
                           runner=args.runner, train_graph=bool(args.train_graph), steps_per_rollout_graph=int(getattr(runner, "_graph_steps", 1)),
                           breakdown_ms=bd))
