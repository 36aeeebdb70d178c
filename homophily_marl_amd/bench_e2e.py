"""End-to-end workload of bench.py: the full homophily training loop on vectorised envs.

One bench "step" = ONE WHOLE ITERATION of the reference's training loop (run.py:181-210) on the N envs of this rank: reset,
`episode_limit` timesteps (observation storage, env-head and incentive-head action selection = Q-net forward, the fused
ssd_step_observe launch), the slot-T bootstrapping pass, replay insertion, sampling and `train_steps_per_rollout`
learner.train calls (reference cadence: one; batch_size 16).  Nothing is skipped inside the timed region and the timed region
holds exactly `steps` rollouts and `steps * train_steps_per_rollout` optimisation steps, whatever the arguments are.
"""
import time

import torch as th
import torch.distributed as dist

from . import abi
from .run import load_config, setup


def _event_time(fn, reps, rounds=3):
    """average / median microseconds of one call of fn, from HIP events (torch's current stream = the launch stream) bracketing
    `rounds` runs of `reps` back-to-back launches."""
    for _ in range(5):
        fn()
    per = []
    for _ in range(rounds):
        s, e = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        th.cuda.synchronize()
        per.append(1e3 * s.elapsed_time(e) / reps)
    per.sort()
    return sum(per) / len(per), per[len(per) // 2]


def controller_flops(c, inp):
    """algorithmic FLOPs (2 x MAC of the reference's f32 layers, homophily_agent.py:20-27,154-208) per agent-step."""
    n, A, V = c["n_agents"], c["n_actions"], 2 * c["view_size"] + 1
    P = (V - 2) * (V - 2)
    E = A + 7
    return dict(encode_conv=2 * 27 * 6 * P, encode_lin=2 * 6 * P * 32,
                head_env=2 * (inp * 64 + 6 * 64 * 64 + 64 * (A + 1)),
                head_inc=2 * ((inp + A) * 64 + 6 * 64 * 64 + n * (64 + E) * 4))


def run_e2e(args, c, rank, world, local_rank):
    N, n, T = c["n_env"], c["n_agents"], 100
    # replay capacity: at least the reference's 5000 episodes (config/default.yaml), rounded up to a multiple of the env batch so
    # that the runner can write its episodes straight into the buffer's slots (ReplayBuffer.reserve: insertion moves no data)
    buffer_size = -(-5000 // N) * N
    spg = max(1, int(getattr(args, "steps_per_graph", 10)))
    while T % spg:
        spg -= 1
    qnet = getattr(args, "qnet_dtype", "fp32")
    tspr = max(1, int(getattr(args, "train_steps_per_rollout", 1)))
    ldt = str(getattr(args, "learner_dtype", "fp32"))
    cfg = load_config(c["env"], overrides=dict(
        runner=args.runner, train_graph=args.train_graph, steps_per_graph=spg, batch_size_run=N, batch_size=16, buffer_size=buffer_size,
        obs_storage=getattr(args, "obs_storage", "f32"), buffer_cpu_only=False, store_state=False, qnet_dtype=qnet,
        train_steps_per_rollout=tspr, learner_dtype=ldt,
        env_args=dict(num_agents=n, map=c["map"], episode_limit=T, view_size=c["view_size"], seed=1), use_cuda=True, save_model=False,
        device_index=local_rank, env_id_base=rank * N, strict_device_ops=True))       # runner / learner statistics on, at the shipped log intervals
    th.manual_seed(0)                     # fixed-seed random-init weights, identical on every rank
    ctx = setup(cfg)
    runner, learner, buf = ctx.runner, ctx.learner, ctx.buffer
    a = ctx.args
    state = dict(episode=0, trains=0, timesteps=0)

    from .run import train_iteration

    def iteration():
        # the shipped driver loop body itself (run.train_iteration = run.py:181-210: runner.run, insert, sample, learner.train)
        before = ctx.train_steps
        state["episode"] = train_iteration(ctx, state["episode"])
        state["timesteps"] += T
        state["trains"] += ctx.train_steps - before

    # Setup (not warm-up): build the hipGraphs.  The rollout graph of a replay slab is captured at the start of the 2nd episode
    # that lands in it and the two train-step graphs at the 3rd learner.train call, so every slab is visited twice before the W
    # warm-up iterations; this is the analogue of compiling the step and is excluded from the warm-up and the timed region.
    for _ in range(max(4, 2 * (buffer_size // N) + 1)):
        iteration()
    from .run import settle_gc
    settle_gc()                           # the shipped loop does the same once its graphs exist (run.run_sequential)
    for _ in range(args.warmup):
        iteration()
    th.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    th.cuda.synchronize()
    state["trains"] = state["timesteps"] = 0
    learner.profile_collectives = bool(getattr(learner, "distributed", False))
    if learner.profile_collectives:
        learner.collective_times()                  # (reset)
    prof = None
    if __import__("os").environ.get("SSD_BENCH_TRACE") == "profile":      # (diagnostic: host-side profile of the timed loop)
        import cProfile
        prof = cProfile.Profile(); prof.enable()
    t0 = time.perf_counter()
    trace = [] if __import__("os").environ.get("SSD_BENCH_TRACE") else None
    if trace is not None:
        ctx._trace_marks = []
    for _ in range(args.steps):
        if trace is not None:
            e0 = th.cuda.Event(enable_timing=True); e0.record()
        iteration()
        if trace is not None:
            e1 = th.cuda.Event(enable_timing=True); e1.record()
            trace.append((1e3 * (time.perf_counter() - t0), e0, e1))      # (diagnostic: HOST time at which the iteration was issued + device events)
    th.cuda.synchronize()
    if prof is not None:
        import pstats
        prof.disable(); pstats.Stats(prof, stream=__import__("sys").stderr).sort_stats("tottime").print_stats(14)
    if trace is not None:
        print("[bench trace] host issue times (ms): " + " ".join("%.1f" % x[0] for x in trace) + " | drained at %.1f" % (1e3 * (time.perf_counter() - t0)), file=__import__("sys").stderr, flush=True)
        print("[bench trace] device ms per iteration: " + " ".join("%.1f" % x[1].elapsed_time(x[2]) for x in trace), file=__import__("sys").stderr, flush=True)
        mk = ctx._trace_marks
        gaps = [(mk[i][0], mk[i - 1][1].elapsed_time(mk[i][1])) for i in range(1, len(mk))]
        slow = [(i, t, round(d, 2)) for i, (t, d) in enumerate(gaps) if (t == "train" and d > 2.0) or (t == "rollout" and d > 8.0) or (t == "start" and d > 1.0)]
        print("[bench trace] phases slower than usual (index, phase ending, device ms): %s" % slow, file=__import__("sys").stderr, flush=True)
        ctx._trace_marks = None
    rank_elapsed = time.perf_counter() - t0          # this rank's own clock, before it waits for the others
    if dist.is_initialized():
        dist.barrier()
    th.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    coll = learner.collective_times() if learner.profile_collectives else None
    learner.profile_collectives = False
    timed = dict(state)
    assert runner.env.native.poll_error() == 0
    assert timed["trains"] == args.steps * tspr and timed["timesteps"] == args.steps * T

    # Steady-state breakdown of one more rollout + train iteration (synchronised between phases; NOT part of `elapsed`)
    def timed_ms(fn):
        th.cuda.synchronize(); t = time.perf_counter(); r = fn(); th.cuda.synchronize()
        return r, 1e3 * (time.perf_counter() - t)
    bd = {}
    _, bd["begin_episode_ms"] = timed_ms(lambda: runner.begin_episode(False))
    _, bd["rollout_100_steps_ms"] = timed_ms(lambda: [runner.step_once() for _ in range(T)])
    batch, bd["finish_episode_ms"] = timed_ms(runner.finish_episode)
    _, bd["replay_insert_ms"] = timed_ms(lambda: buf.insert_episode_batch(batch))
    sample, bd["sample_ms"] = timed_ms(lambda: buf.sample(a.batch_size, out=learner.sample_out())[:, :T + 1])      # as run.train_iteration calls it
    _, bd["learner_train_ms"] = timed_ms(lambda: learner.train(sample, runner.t_env, ctx.train_steps if a.schedule_unit == "rollouts" else state["episode"]))
    bd = {k: round(v, 3) for k, v in bd.items()}

    # Kernel timing for the roofline objects.  Inside a hipGraph replay there is no host call to bracket, so the SAME launches on
    # the SAME live objects (runner.timestep_launches(): the closures the graph was captured from) are timed right after the timed
    # region with HIP events around runs of back-to-back launches of one kernel (every launch is idempotent in its cost: the data it
    # reads is whatever the rollout left).  The env kernel is timed over three whole episodes (reset, then T launches).
    from bench import MFMA_BF16_PEAK_TF, MFMA_F32_PEAK_TF, algorithmic_bytes_per_env_step
    code = getattr(args, "obs_storage", "f32") == "code"
    V = 2 * c["view_size"] + 1
    env_bytes = algorithmic_bytes_per_env_step(c["H"], c["W"], n, V, 1, 1) if code else algorithmic_bytes_per_env_step(c["H"], c["W"], n, V)
    if not code and "code" in getattr(runner, "cur", {}):
        env_bytes += n * V * V                      # f32 storage: plus the u8 channel-mask window the fused encoder reads
    env_bytes *= N
    kernels = []
    fl = controller_flops(c, ctx.mac.input_shape)
    nprod = getattr(runner.fast, "n_products", None) if getattr(runner, "fast", None) is not None else None
    if hasattr(runner, "timestep_launches"):
        runner.begin_episode(False)
        for _ in range(10):
            runner.step_once()                      # a live mid-episode state
        th.cuda.synchronize()
        for name, key, fn in runner.timestep_launches():
            if key == "env":
                continue
            avg, med = _event_time(fn, 50)
            # pipelined rollout: the inc head of t and the encoder of t + 1 are one launch (ssd::k_inc_encode)
            parts = {"encode": ["encode_conv", "encode_lin"], "inc_encode": ["head_inc", "encode_conv", "encode_lin"]}.get(key, [key])
            alg = sum(fl[p] for p in parts) * N * n
            k = dict(name=name, avg_us=avg, median_us=med, bound="mfma", flops_per_launch=alg)
            if nprod:
                # precision 2: an f32 product is evaluated as 3 f16 MFMA products of two-term splits (the one-hot conv planes are
                # exact: 2) -- count the 16-bit MFMA work the f32-equivalent result needs against the dense 16-bit MFMA peak
                issued = sum(fl[p] * nprod[p] for p in parts) * N * n
                k.update(flops_per_launch=issued, peak_tf=MFMA_BF16_PEAK_TF,
                         mfma_dtype=("f16 two-term splits, f32 accumulate (f32-equivalent)" if max(nprod.values()) > 1 else "bf16, f32 accumulate"),
                         note="algorithmic f32 FLOPs %d; 16-bit MFMA products per f32 product: %s%s" % (
                             alg, {p: nprod[p] for p in parts},
                             "; encode_conv 0 = the convolution is a class-LUT table sum on the vector unit (exact f32, no matrix-core "
                             "products): its FLOPs are algorithmic but not part of the issued MFMA work priced here" if nprod.get("encode_conv", 1) == 0 and "encode_conv" in parts else ""))
                if max(nprod.values()) > 1:
                    # the same launch priced the way round 1's f32 kernels were: the reference's f32 FLOPs against the f32-MFMA peak
                    k["algorithmic_f32"] = dict(flops_per_launch=alg, peak_tf=MFMA_F32_PEAK_TF)
            kernels.append(k)
    env = runner.env
    avail = th.nonzero(env.avail_actions_batch[0, 0]).squeeze(-1).to(th.int32)
    acts = [avail[th.randint(0, avail.numel(), (N, n), device=env.device)].contiguous() for _ in range(8)]
    # the env launch exactly as the rollout issues it (_fast_stages.env_step): observation into slot ep_step of the episode storage
    # (fresh HBM lines every launch -- a dense buffer rewritten in place would be absorbed by the 256 MiB Infinity Cache), plus the
    # channel-mask side output when the encoder reads one
    out = runner.cur if getattr(runner, "direct_obs", False) else None
    per = []
    for rep in range(4):                          # whole episodes: reset, then T back-to-back launches (the first one is un-timed warm-up)
        runner.begin_episode(False)
        s, e = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        s.record()
        for i in range(T):
            env.step_batch(acts[i % 8], observe=True, fmt=runner.obs_fmt, out=out)
        e.record()
        th.cuda.synchronize()
        if rep:
            per.append(1e3 * s.elapsed_time(e) / T)
    assert env.native.poll_error() == 0
    per.sort()
    kernels.append(dict(name="ssd::k_env<MODE_STEP_OBS>", avg_us=sum(per) / len(per), median_us=per[len(per) // 2], bound="hbm",
                        bytes_per_launch=env_bytes))
    tot = sum(k["avg_us"] for k in kernels)
    for k in kernels:
        k["share_of_timestep"] = round(k["avg_us"] / tot, 4)
    return dict(elapsed=elapsed, rank_elapsed=rank_elapsed, collectives=coll, grad_bytes=4 * sum(p.numel() for p in learner.params), kernels=kernels,
                dtype="fp32" if (qnet == "fp32" and ldt == "fp32") else ("bf16" if (qnet != "fp32" and ldt != "fp32") else "mixed: rollout %s, learner %s" % (qnet, ldt)),
                workload="%s_rollout_plus_homophily_train" % args.config,
                extra=dict(obs_format=("u8 class codes [n_env,n,%d,%d] (format C)" % (V, V)) if code else "f32[n_env,n,3,%d,%d]" % (V, V),
                           qnet_dtype=("fp32 (rollout: two-term f16 split MFMA products, f32-equivalent)" if qnet == "fp32"
                                       else "bf16 rollout inference (single bf16 MFMA products)"),
                           learner_dtype=("fp32 (exact-f32 MFMAs in the affine layers, f32-equivalent split products in the recurrence and the encoder)" if ldt == "fp32"
                                          else "bf16 (LABELLED VARIANT: single bf16 MFMA products in the affine layers, the recurrence and the encoder; f32 master weights, Adam and loss)"),
                           step="1 bench step = 1 iteration: reset + %d timesteps + slot-T pass + replay insert + sample + %d learner.train" % (T, tspr),
                           timesteps_timed=timed["timesteps"], train_steps_timed=timed["trains"], rollouts_timed=args.steps,
                           train="learner.train(batch_size 16 x T 101), double-Q + sim loss, 2x Adam; %d per rollout" % tspr,
                           buffer="device-resident ReplayBuffer, %d episodes, %s" % (buf.buffer_size, "written in place by the runner" if getattr(runner, "_replay", None) is not None else "copy insertion"),
                           runner=args.runner, train_graph=bool(args.train_graph), steps_per_rollout_graph=int(getattr(runner, "_graph_steps", 1)),
                           breakdown_ms=bd))
