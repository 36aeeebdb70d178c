#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the SSD hot path on N MI355X GPUs (one process per GPU), BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W [--workload env|e2e] [--n-env 4096]
    (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): Cleanup `default5`, 5 agents, 4096 vectorised envs per GPU, episode_limit 100,
default extra_args, COUNTER-mode env RNG (seed 1), synthetic actions.
  --workload env : one "step" = one transition of all envs: the fused ssd_step_observe launch (moves, beams, respawn,
                   rewards, egocentric fp32 obs of the new state) with actions i.i.d. uniform over the available set,
                   pre-generated in HBM; ssd_reset at every episode boundary is inside the timed region.
  --workload e2e : one "step" = one transition of all envs inside the full loop: Q-net action selection (env + incentive
                   heads), ssd_step_observe, rollout storage, and one learner.train per 100-step rollout.
Rank 0 prints ONE JSON line with the driver's fields plus `roofline` (dominant kernel k_env<STEP_OBS>, HIP-event timed
on the launch stream) and `cpu_baseline` (the C oracle on the host cores, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured streaming ceiling


def algorithmic_bytes_per_env_step(H, W, n, V, obs_bytes_per_elem=4, planes=3):
    """grid r+w, actions, agent state r+w, obs, 7 f32 scalars per agent.  Format R (default): obs f32
    [n, 3, V, V]; format C: planes = 1, obs_bytes_per_elem = 1 (u8 class codes [n, V, V])."""
    return 2 * H * W + 4 * n + 16 * n + planes * obs_bytes_per_elem * n * V * V + 28 * n


def cpu_baseline(n_env, steps):
    """The CPU restatement (oracle/, kind "port") on the host cores: same workload, bounded sample.  The serial C port is run as
    one env shard per thread (ctypes releases the GIL; envs never interact, so this is the CPU's own data-parallel form); the
    single-thread rate of a shorter sample is reported in `sample`."""
    import threading
    import numpy as np
    from homophily_marl_amd import abi
    from oracle.oracle_py import OracleEnv
    n = 5
    avail = np.array([0, 1, 2, 3, 4, 8])

    def run(shard_envs, env_id_base, nsteps, out, k):
        env = OracleEnv("cleanup", map="default5", num_agents=n, n_env=shard_envs, view_size=7, episode_limit=100,
                        rng_mode=abi.RNG_COUNTER, seed=1, env_id_base=env_id_base)
        rng = np.random.default_rng(0x5D5D + k)
        acts = [avail[rng.integers(0, 6, (shard_envs, n))].astype(np.int32) for _ in range(16)]
        env.reset()
        t0 = time.perf_counter()
        for t in range(nsteps):
            if t and t % 100 == 0:
                env.reset()
            env.step(acts[t % 16])
            env.observe(abi.OBS_F32)
        out[k] = time.perf_counter() - t0

    one = [0.0]
    s1 = max(100, steps // 4)
    run(n_env, 0, s1, one, 0)                                  # single thread, all envs
    cores = max(1, min(16, os.cpu_count() or 1, n_env))
    shard = n_env // cores
    dts = [0.0] * cores
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(shard, k * shard, steps, dts, k)) for k in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    return dict(value=shard * cores * n * steps / max(dts), unit="agent-steps/s", cores=cores, kind="port",
                sample="%d envs x %d steps of step+observe(fp32), C oracle, one shard of %d envs per thread, %.1f s; "
                       "single thread: %.0f agent-steps/s (%d steps, %.1f s)" % (shard * cores, steps, shard, wall,
                                                                                n_env * n * s1 / one[0], s1, one[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--n-env", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", default=os.environ.get("SSD_BENCH_WORKLOAD", "e2e"), choices=["env", "e2e"])
    ap.add_argument("--runner", default="hip_graph", choices=["hip_vec", "hip_graph"], help="e2e: rollout runner")
    ap.add_argument("--train-graph", type=int, default=1, help="e2e: capture the train step as hipGraphs")
    ap.add_argument("--steps-per-graph", type=int, default=10, help="e2e: timesteps captured per rollout hipGraph")
    ap.add_argument("--obs-storage", default="f32", choices=["f32", "code"],
                    help="e2e: observation format of the episode storage / replay buffer: f32 planes (format R, the reference's) or "
                         "u8 class codes (format C: 12x fewer observation bytes; the roofline object then uses format C bytes)")
    ap.add_argument("--warm", type=float, default=0.0,
                    help="env workload: fraction of the waste cells turned into clean river after every reset (SURVEY.md 8d 'warm' variant: "
                         "exercises apple spawning; 0 = start from the map's reset state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-steps", type=int, default=2000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from homophily_marl_amd import abi
    from homophily_marl_amd.envs.native import NativeEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        backend = os.environ.get("SSD_DIST_BACKEND", "nccl")     # "nccl" = RCCL; "gloo" only for rehearsing >1 rank on one GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n, N, T = 5, args.n_env, 100
    if args.workload == "e2e":
        from homophily_marl_amd.bench_e2e import run_e2e
        result = run_e2e(args, rank, world, local_rank)
    else:
        env = NativeEnv("cleanup", device=local_rank, map="default5", num_agents=n, n_env=N, view_size=7, episode_limit=T,
                        rng_mode=abi.RNG_COUNTER, seed=1, env_id_base=rank * N)
        # synthetic actions: i.i.d. uniform over the available set {0,1,2,3,4,8}, resident in HBM
        g = torch.Generator(device=dev).manual_seed(0x5D5D + rank)
        avail = torch.tensor([0, 1, 2, 3, 4, 8], dtype=torch.int32, device=dev)
        n_act = 64
        acts = [avail[torch.randint(0, 6, (N, n), generator=g, device=dev)].contiguous() for _ in range(n_act)]
        bufs = env.obs_buffers(abi.OBS_F32)

        def reset_env():
            env.reset()
            if args.warm > 0:      # pre-clean part of the waste through the state import (not a kernel of the path; outside the brackets)
                grid = env.export_state()["grid"]
                hit = (grid == 3) & (torch.rand(grid.shape, generator=g, device=dev) < args.warm)
                env.import_state(grid=torch.where(hit, torch.full_like(grid, 4), grid))

        def one_step(t):
            if t % T == 0:
                reset_env()
            env.step_observe(acts[t % n_act], out=bufs)

        for t in range(args.warmup):
            one_step(t)
        # kernel time for the roofline: HIP events (torch's current stream = the launch stream) bracketing every run of
        # back-to-back k_env<STEP_OBS> launches between two resets; average = bracket time / launches in it
        ev, run_len = [], []
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        open_ev = None
        for t in range(args.steps):
            if (args.warmup + t) % T == 0:
                if open_ev is not None:
                    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((open_ev, e)); open_ev = None
                reset_env()
            if open_ev is None:
                open_ev = torch.cuda.Event(enable_timing=True); open_ev.record(); run_len.append(0)
            env.step_observe(acts[t % n_act], out=bufs)
            run_len[-1] += 1
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((open_ev, e))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        assert env.poll_error() == 0
        per_launch = sorted(1e3 * a.elapsed_time(b) / k for (a, b), k in zip(ev, run_len))
        kern_avg_us = sum(1e3 * a.elapsed_time(b) for a, b in ev) / sum(run_len)
        kern_med_us = per_launch[len(per_launch) // 2]
        bytes_per_launch = algorithmic_bytes_per_env_step(env.H, env.W, n, env.V) * N
        result = dict(elapsed=elapsed, kern_avg_us=kern_avg_us, kern_med_us=kern_med_us, bytes_per_launch=bytes_per_launch,
                      dtype="u8", workload="cleanup_default5_env_step_observe_fp32obs" + ("_warm%d" % round(100 * args.warm) if args.warm > 0 else ""),
                      extra=dict(obs_format="f32[n_env,n,3,15,15]", kernel="ssd::k_env<MODE_STEP_OBS>"))

    elapsed = result["elapsed"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        total_agent_steps = N * n * args.steps * world
        achieved = result["bytes_per_launch"] / (result["kern_avg_us"] * 1e-6) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            tr = json.load(open(tj))
            if tr.get("kernel") == result["extra"]["kernel"] and tr.get("n_env") == N and result["extra"]["obs_format"].startswith("f32"):
                traffic = tr.get("hbm_bytes_per_launch")
        line = {
            "metric": "agent_steps_per_sec", "value": total_agent_steps / elapsed, "unit": "agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": result["dtype"], "data": "synthetic",
            "config": dict({"workload": result["workload"], "env": "cleanup", "map": "default5", "n_agents": n,
                            "n_env_per_gpu": N, "episode_limit": T, "rng": "counter(philox4x32-10, seed 1)",
                            "parallelism": ("dp%d: env shards, no collective in the rollout" % world) +
                                           ("; 1 flat-gradient all-reduce + 2 scalars per train step" if args.workload == "e2e" else "")},
                           **result["extra"]),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_avg_us": result["kern_avg_us"], "kernel_median_us": result["kern_med_us"],
                         "algorithmic_bytes_per_launch": result["bytes_per_launch"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_sample_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
