#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the SSD hot path on N MI355X GPUs (one process per GPU), BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W [--config cleanup5|harvest5|cleanup10] [--workload e2e|env]
    N > 1 works both ways: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env), and as plain `python bench.py --gpus N ...`: without WORLD_SIZE in
    the env the process starts that launcher itself as a CHILD (before anything touches the GPU; never an exec), relays rank 0's
    one JSON line and exits with the launcher's return code.  `--launch-dry-run` takes the same front door with gloo ranks that
    only form the group and count each other (no GPU; the CPU suite runs it).

Configurations (BASELINE.json `configs`; episode_limit 100, default extra_args, COUNTER-mode env RNG seed 1):
  cleanup5  (default, configs[1], the one the metric is quoted on): Cleanup `default5`, 5 agents, 4096 envs per GPU
  harvest5  (configs[2]): Harvest `default10` map, view_size 15 (31 x 31 windows), 5 agents, 4096 envs per GPU
  cleanup10 (configs[3]): Cleanup `default10`, 10 agents, 8192 envs per GPU
Workloads:
  e2e (default): ONE STEP = ONE WHOLE TRAINING ITERATION of the reference loop (run.py:181-210) on all envs of the rank:
        reset, episode_limit timesteps (Q-net action selection of both heads, fused ssd_step_observe, storage), the slot-T
        bootstrapping pass, replay insertion, sampling and learner.train (batch 16 x T 101, double-Q + similarity loss, 2x Adam).
        value = n_env * n_agents * episode_limit * steps * world / elapsed.  Nothing is skipped in the timed region, whatever
        --steps / --warmup are.
  env: one step = one transition of all envs: the fused ssd_step_observe launch with pre-generated synthetic actions
        (resets at the episode boundaries are inside the timed region).
Rank 0 prints ONE JSON line with the driver's fields plus
  `roofline`     : the kernel with the largest share of the timestep (measured live with HIP events on the launch stream),
  `kernels`      : the same object for every kernel of the rollout timestep (e2e),
  `cpu_baseline` : the C oracle on the host cores (rank 0, N = 1 only),
  `env_workload` : (e2e, N = 1) the SURVEY 8(d) kernel-only workload -- fused step + observe with f32 observations -- run for 300
                   transitions in the same process AFTER the timed region: the HBM fraction of the path's byte-moving kernel,
  `config.per_rank` : (N > 1) every rank's own ms_per_step and the time of its gradient all-reduce.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured streaming ceiling
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
MFMA_F32_PEAK_TF = 157.3    # f32-input MFMA = f32 vector peak

CONFIGS = {
    "cleanup5": dict(env="cleanup", map="default5", n_agents=5, view_size=7, n_env=4096, H=25, W=18, n_actions=9),
    "harvest5": dict(env="harvest", map="default10", n_agents=5, view_size=15, n_env=4096, H=9, W=38, n_actions=8),
    "cleanup10": dict(env="cleanup", map="default10", n_agents=10, view_size=7, n_env=8192, H=48, W=18, n_actions=9),
}


def algorithmic_bytes_per_env_step(H, W, n, V, obs_bytes_per_elem=4, planes=3):
    """grid r+w, actions, agent state r+w, obs, 7 f32 scalars per agent.  Format R (default): obs f32
    [n, 3, V, V]; format C: planes = 1, obs_bytes_per_elem = 1 (u8 class codes [n, V, V])."""
    return 2 * H * W + 4 * n + 16 * n + planes * obs_bytes_per_elem * n * V * V + 28 * n


def host_threads():
    """threads the CPU baseline may use: the scheduler affinity of this process (the GPU box gives a 1-GPU job 16), capped at 32."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(int(os.environ.get("SSD_CPU_THREADS", "32")), avail))


def cpu_baseline(c, n_env, steps):
    """The CPU restatement (oracle/, kind "port") on the host cores: same workload, bounded sample.  The serial C port is run as
    one env shard per thread (ctypes releases the GIL; envs never interact, so this is the CPU's own data-parallel form); the
    single-thread rate of a shorter sample is reported in `sample`."""
    import threading
    import numpy as np
    from homophily_marl_amd import abi
    from oracle.oracle_py import OracleEnv
    n = c["n_agents"]
    avail = np.array([0, 1, 2, 3, 4, 8] if c["env"] == "cleanup" else [0, 1, 2, 3, 4])

    def run(shard_envs, env_id_base, nsteps, out, k):
        env = OracleEnv(c["env"], map=c["map"], num_agents=n, n_env=shard_envs, view_size=c["view_size"], episode_limit=100,
                        rng_mode=abi.RNG_COUNTER, seed=1, env_id_base=env_id_base)
        rng = np.random.default_rng(0x5D5D + k)
        acts = [avail[rng.integers(0, len(avail), (shard_envs, n))].astype(np.int32) for _ in range(16)]
        env.reset()
        t0 = time.perf_counter()
        for t in range(nsteps):
            if t and t % 100 == 0:
                env.reset()
            env.step(acts[t % 16])
            env.observe(abi.OBS_F32)
        out[k] = time.perf_counter() - t0

    one = [0.0]
    s1 = max(50, steps // 8)
    run(n_env, 0, s1, one, 0)                                  # single thread, all envs
    cores = min(host_threads(), n_env)
    shard = n_env // cores
    dts = [0.0] * cores
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(shard, k * shard, steps, dts, k)) for k in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    return dict(value=shard * cores * n * steps / max(dts), unit="agent-steps/s", cores=cores, host_cpus=os.cpu_count(), kind="port",
                sample="%d envs x %d steps of step+observe(fp32), C oracle, one shard of %d envs per thread, %.1f s; "
                       "single thread: %.0f agent-steps/s (%d steps, %.1f s)" % (shard * cores, steps, shard, wall,
                                                                                n_env * n * s1 / one[0], s1, one[0]))


def roofline_entry(k):
    """k: dict(name, avg_us, median_us, bound, and bytes_per_launch | flops_per_launch [+ issued_flops_per_launch, peak_tf])."""
    if k["bound"] == "hbm":
        achieved = k["bytes_per_launch"] / (k["avg_us"] * 1e-6) / 1e9
        e = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "traffic": k.get("traffic"), "algorithmic_bytes_per_launch": k["bytes_per_launch"]}
    else:
        achieved = k["flops_per_launch"] / (k["avg_us"] * 1e-6) / 1e12
        peak = k.get("peak_tf", MFMA_F32_PEAK_TF)
        e = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
             "traffic": k.get("traffic"), "algorithmic_flops_per_launch": k["flops_per_launch"]}
        for key in ("mfma_dtype", "note"):
            if key in k:
                e[key] = k[key]
        if "algorithmic_f32" in k:      # informational: reference-f32 FLOPs / duration against the f32-MFMA (= f32 vector) peak
            a32 = k["algorithmic_f32"]
            ach = a32["flops_per_launch"] / (k["avg_us"] * 1e-6) / 1e12
            e["algorithmic_f32"] = {"achieved": ach, "peak": a32["peak_tf"], "unit": "TFLOP/s", "frac": ach / a32["peak_tf"],
                                    "algorithmic_flops_per_launch": a32["flops_per_launch"]}
    e["traffic_source"] = k.get("traffic_source")
    e.update(kernel=k["name"], kernel_avg_us=k["avg_us"], kernel_median_us=k.get("median_us", k["avg_us"]))
    if "share_of_timestep" in k:
        e["share_of_timestep"] = k["share_of_timestep"]
    return e


def run_env_workload(args, c, rank, local_rank, steps, warmup, in_group=False):
    """SURVEY.md section 8(d)'s kernel-only workload: the fused ssd_step_observe launch on pre-generated synthetic actions, format R
    (f32 observation [n_env, n, 3, V, V]); resets at the episode boundaries are inside the wall-clock bracket, the kernel time comes
    from HIP events around every run of back-to-back launches between two resets."""
    import torch
    import torch.distributed as dist
    from homophily_marl_amd import abi
    from homophily_marl_amd.envs.native import NativeEnv
    n, N, T, V = c["n_agents"], c["n_env"], 100, 2 * c["view_size"] + 1
    dev = torch.device("cuda", local_rank)
    env = NativeEnv(c["env"], device=local_rank, map=c["map"], num_agents=n, n_env=N, view_size=c["view_size"], episode_limit=T,
                    rng_mode=abi.RNG_COUNTER, seed=1, env_id_base=rank * N)
    # synthetic actions: i.i.d. uniform over the available set, resident in HBM
    g = torch.Generator(device=dev).manual_seed(0x5D5D + rank)
    avail = torch.tensor([0, 1, 2, 3, 4, 8] if c["env"] == "cleanup" else [0, 1, 2, 3, 4], dtype=torch.int32, device=dev)
    n_act = 64
    acts = [avail[torch.randint(0, avail.numel(), (N, n), generator=g, device=dev)].contiguous() for _ in range(n_act)]
    bufs = env.obs_buffers(abi.OBS_F32)

    def reset_env():
        env.reset()
        if args.warm > 0:      # pre-clean part of the waste through the state import (not a kernel of the path; outside the brackets)
            grid = env.export_state()["grid"]
            hit = (grid == 3) & (torch.rand(grid.shape, generator=g, device=dev) < args.warm)
            env.import_state(grid=torch.where(hit, torch.full_like(grid, 4), grid))

    for t in range(warmup):
        if t % T == 0:
            reset_env()
        env.step_observe(acts[t % n_act], out=bufs)
    # kernel time for the roofline: HIP events (torch's current stream = the launch stream) bracketing every run of
    # back-to-back k_env<STEP_OBS> launches between two resets; average = bracket time / launches in it
    ev, run_len = [], []
    torch.cuda.synchronize()
    if in_group:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    open_ev = None
    for t in range(steps):
        if (warmup + t) % T == 0:
            if open_ev is not None:
                e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((open_ev, e)); open_ev = None
            reset_env()
        if open_ev is None:
            open_ev = torch.cuda.Event(enable_timing=True); open_ev.record(); run_len.append(0)
        env.step_observe(acts[t % n_act], out=bufs)
        run_len[-1] += 1
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((open_ev, e))
    torch.cuda.synchronize()
    if in_group:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert env.poll_error() == 0
    per_launch = sorted(1e3 * a.elapsed_time(b) / k for (a, b), k in zip(ev, run_len))
    kern = dict(name="ssd::k_env<MODE_STEP_OBS>", bound="hbm", avg_us=sum(1e3 * a.elapsed_time(b) for a, b in ev) / sum(run_len),
                median_us=per_launch[len(per_launch) // 2], bytes_per_launch=algorithmic_bytes_per_env_step(env.H, env.W, n, env.V) * N)
    return dict(elapsed=elapsed, kernels=[kern], dtype="u8",
                workload="%s_env_step_observe_fp32obs" % args.config + ("_warm%d" % round(100 * args.warm) if args.warm > 0 else ""),
                extra=dict(obs_format="f32[n_env,n,3,%d,%d]" % (V, V)))


def attach_traffic(kernels, config, suffix, N):
    """PMC traffic of the committed rocprofv3 --pmc passes, per launch (same kernel, same sizes only)"""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tj):
        return
    tr = json.load(open(tj))
    for k in kernels:
        rec = tr.get("kernels", {}).get("%s@%s%s" % (k["name"], config, suffix))
        if rec and rec.get("n_env") == N:
            k["traffic"] = rec.get("hbm_bytes_per_launch")
            k["traffic_source"] = "profiles/traffic.json (committed rocprofv3 --pmc passes of %s; not measured in this run)" % tr.get("collected", "an earlier collection")


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child `torch.distributed.run` job (one process per GPU),
    pass rank 0's JSON line through and return the job's exit code.  Called before torch is imported: this process never
    initialises the GPU and never execs."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SSD_SELF_LAUNCHED="1")
    print("[bench] self-launch: %s" % " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    js = []
    for raw in p.stdout:                                   # relayed as the ranks print: everything but JSON lines goes to stderr at once
        l = raw.decode(errors="replace").rstrip("\n")
        if not l.strip():
            continue
        if l.lstrip().startswith("{"):
            js.append(l)
        else:
            print(l, file=sys.stderr, flush=True)
    rc = p.wait()
    for l in js[:-1]:                                      # (only the last JSON line is the job's result)
        print(l, file=sys.stderr)
    if rc == 0 and not js:
        print("[bench] the ranks exited 0 without a JSON line", file=sys.stderr)
        rc = 1
    if js and rc == 0:
        sys.stdout.write(js[-1] + "\n")
        sys.stdout.flush()
    return rc


def dry_run(args):
    """--launch-dry-run: the ranks form a gloo group, count each other with an all-reduce of ones and rank 0 prints a line with
    the launch-related fields only (no GPU, no workload)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen = 1
    if os.environ.get("SSD_DRY_RUN_FAIL_RANK") == str(rank):      # test hook: a rank that dies must fail the whole front door
        raise SystemExit(7)
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
    if rank == 0:
        print(json.dumps({"metric": "agent_steps_per_sec", "value": None, "n_gpus": world, "dry_run": True,
                          "config": {"world_size": world, "ranks_seen": seen, "backend": "gloo" if world > 1 else "none",
                                     "self_launched": os.environ.get("SSD_SELF_LAUNCHED") == "1"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if seen == world == args.gpus else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="e2e: whole iterations (default 20); env: transitions (default 500)")
    ap.add_argument("--warmup", type=int, default=None, help="e2e: whole iterations (default 5); env: transitions (default 100)")
    ap.add_argument("--config", default=os.environ.get("SSD_BENCH_CONFIG", "cleanup5"), choices=sorted(CONFIGS))
    ap.add_argument("--n-env", type=int, default=None, help="envs per GPU (default: the configuration's)")
    ap.add_argument("--workload", default=os.environ.get("SSD_BENCH_WORKLOAD", "e2e"), choices=["env", "e2e"])
    ap.add_argument("--runner", default="hip_graph", choices=["hip_vec", "hip_graph"], help="e2e: rollout runner")
    ap.add_argument("--train-graph", type=int, default=1, help="e2e: capture the train step as hipGraphs")
    ap.add_argument("--steps-per-graph", type=int, default=10, help="e2e: timesteps captured per rollout hipGraph")
    ap.add_argument("--qnet-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="e2e: arithmetic of the ROLLOUT controller kernels: fp32 (two-term f16 split MFMA products, f32-equivalent; the headline) "
                         "or bf16 (single bf16 products; a second, labelled line -- the learner stays fp32)")
    ap.add_argument("--learner-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="e2e: arithmetic of the LEARNER's matrix products: fp32 (exact-f32 / f32-equivalent split MFMA products; the headline) or "
                         "bf16 (single bf16 MFMA products in the affine layers, the recurrence and the encoder; f32 master weights / Adam / loss) "
                         "-- a labelled line, never the headline")
    ap.add_argument("--train-steps-per-rollout", type=int, default=1, help="e2e: learner.train calls per rollout (reference cadence: 1)")
    ap.add_argument("--obs-storage", default="code", choices=["f32", "code"],
                    help="e2e: observation format of the episode storage / replay buffer: u8 class codes (format C, default: lossless, "
                         "12x fewer observation bytes, decoded bit-exactly where the learner reads the sample) or f32 planes (format R, "
                         "the reference's in-memory layout); the env kernel's roofline entry uses the bytes of the format it emits")
    ap.add_argument("--warm", type=float, default=0.0,
                    help="env workload: fraction of the waste cells turned into clean river after every reset (SURVEY.md 8d 'warm' variant: "
                         "exercises apple spawning; 0 = start from the map's reset state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-env-workload", action="store_true", help="e2e, 1 GPU: skip the format-R env workload measured after the timed region")
    ap.add_argument("--env-workload-steps", type=int, default=300)
    ap.add_argument("--cpu-sample-steps", type=int, default=None)
    ap.add_argument("--launch-dry-run", action="store_true", help="form the process group (gloo), count the ranks, print the launch fields; no GPU")
    args = ap.parse_args()
    # SSD_FORCE_DIST=1 (rehearsal of the collective path on a one-GPU box): a 1-rank job goes through the same front door
    if (args.gpus > 1 or os.environ.get("SSD_FORCE_DIST") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.launch_dry_run:
        sys.exit(dry_run(args))
    # stdout carries exactly ONE line, the JSON: libraries that print banners to fd 1 (RCCL's version header at communicator
    # creation) are sent to stderr for the lifetime of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    c = dict(CONFIGS[args.config])
    if args.n_env:
        c["n_env"] = args.n_env
    if args.steps is None:
        args.steps = 20 if args.workload == "e2e" else 500
    if args.warmup is None:
        args.warmup = 5 if args.workload == "e2e" else 100

    import torch
    import torch.distributed as dist
    from homophily_marl_amd import abi
    from homophily_marl_amd.envs.native import NativeEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "none"
    force_dist = os.environ.get("SSD_FORCE_DIST") == "1"      # rehearsal: a 1-rank process group still runs every collective
    if args.gpus > 1 or world > 1 or force_dist:
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` or torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
        backend = os.environ.get("SSD_DIST_BACKEND", "nccl")     # "nccl" = RCCL; "gloo" only for rehearsing >1 rank on one GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            dist.init_process_group(backend)
    in_group = dist.is_initialized()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ranks_seen = 1
    if in_group:       # how many ranks the collective really joins: an all-reduce of ones on the job's own backend
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, "process group joins %d ranks, WORLD_SIZE is %d" % (ranks_seen, world)

    n, N, T, V = c["n_agents"], c["n_env"], 100, 2 * c["view_size"] + 1
    if args.workload == "e2e":
        from homophily_marl_amd.bench_e2e import run_e2e
        result = run_e2e(args, c, rank, world, local_rank)
        units = N * n * T * args.steps * world
    else:
        result = run_env_workload(args, c, rank, local_rank, args.steps, args.warmup, in_group)
        units = N * n * args.steps * world

    elapsed = result["elapsed"]
    per_rank = None
    if in_group:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own clock (before the closing barrier) and its gradient all-reduce times: a scaling line explains itself
        coll = result.get("collectives") or {}
        mine = torch.tensor([result.get("rank_elapsed", result["elapsed"]), coll.get("device_ms_avg", float("nan")),
                             coll.get("device_ms_max", float("nan")), coll.get("host_ms_avg", float("nan")), float(coll.get("calls", 0))],
                            dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = torch.stack(allr).cpu()
        ms = 1e3 * rows[:, 0] / args.steps
        per_rank = {"ms_per_step_min": float(ms.min()), "ms_per_step_max": float(ms.max()), "ms_per_step": [round(float(x), 4) for x in ms]}
        if float(rows[0, 4]) > 0:
            per_rank["grad_all_reduce"] = {"calls_timed": int(rows[0, 4]), "bytes": result.get("grad_bytes"),
                                           "device_ms_avg_per_rank": [round(float(x), 4) for x in rows[:, 1]],
                                           "device_ms_max_per_rank": [round(float(x), 4) for x in rows[:, 2]],
                                           "host_ms_avg_per_rank": [round(float(x), 4) for x in rows[:, 3]],
                                           "how": "HIP events on the launch stream around dist.all_reduce of the flat gradient (between the two train graphs)"}
    env_wl = None
    if rank == 0 and world == 1 and args.workload == "e2e" and not args.no_env_workload:
        # the HBM claim of the path (step + observe in format R) measured in the SAME process right after the e2e
        # timed region -- outside `elapsed`, the headline value is untouched
        ew = run_env_workload(args, c, rank, local_rank, args.env_workload_steps, 100)
        attach_traffic(ew["kernels"], args.config, "@env_workload", N)
        e = roofline_entry(ew["kernels"][0])
        env_wl = {"kernel": e["kernel"], "workload": ew["workload"], "transitions": args.env_workload_steps,
                  "avg_us": e["kernel_avg_us"], "median_us": e["kernel_median_us"],
                  "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"], "achieved": e["achieved"], "unit": "GB/s",
                  "peak": e["peak"], "frac": e["frac"], "traffic": e["traffic"], "traffic_source": e["traffic_source"],
                  "agent_steps_per_sec": N * n * args.env_workload_steps / ew["elapsed"]}
    if rank == 0:
        kernels = result["kernels"]
        attach_traffic(kernels, args.config, "@env_workload" if args.workload == "env" else ("@f32storage" if args.obs_storage == "f32" else ""), N)
        entries = [roofline_entry(k) for k in kernels]
        dominant = max(entries, key=lambda e: e["kernel_avg_us"])
        line = {
            "metric": "agent_steps_per_sec", "value": units / elapsed, "unit": "agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": result["dtype"], "data": "synthetic",
            "config": dict({"workload": result["workload"], "name": args.config, "env": c["env"], "map": c["map"], "n_agents": n,
                            "n_env_per_gpu": N, "episode_limit": T, "view_size": c["view_size"], "rng": "counter(philox4x32-10, seed 1)",
                            "world_size": world, "ranks_seen": ranks_seen, "backend": {"nccl": "nccl(RCCL)"}.get(backend, backend),
                            "self_launched": os.environ.get("SSD_SELF_LAUNCHED") == "1",
                            "parallelism": ("dp%d: env shards, no collective in the rollout" % world) +
                                           ("; 1 flat-gradient all-reduce + 2 scalars per train step" if args.workload == "e2e" else "")},
                           **result["extra"]),
            "roofline": dominant,
        }
        if len(entries) > 1:
            line["kernels"] = entries
        if env_wl is not None:
            line["env_workload"] = env_wl
        if per_rank is not None:
            line["config"]["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            per_step = {"cleanup5": 2000, "harvest5": 600, "cleanup10": 500}[args.config]
            line["cpu_baseline"] = cpu_baseline(c, N, args.cpu_sample_steps or per_step)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if in_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
