"""Training driver: the reference's run_sequential loop (src/run.py:81-244) re-stated around the same four registries.
One learner.train per runner.run (run.py:184-210).  `load_config` reproduces the yaml layering default <- env <- alg
of src/main.py:57-63,78-90 (with yaml.safe_load); sacred / tensorboard plumbing is out of scope.
"""
import os
from types import SimpleNamespace

import torch as th
import yaml

from .components.episode_buffer import ReplayBuffer
from .components.transforms import OneHot
from .controllers import REGISTRY as mac_REGISTRY
from .learners import REGISTRY as le_REGISTRY
from .runners import REGISTRY as r_REGISTRY
from .utils.logging import Logger

_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


def _merge(d, u):
    for k, v in u.items():
        d[k] = _merge(d.get(k, {}), v) if isinstance(v, dict) else v
    return d


def load_config(env_config="cleanup", alg_config="homophily", overrides=None):
    cfg = {}
    for name in ("default", env_config, alg_config):
        with open(os.path.join(_CFG, name + ".yaml")) as f:
            _merge(cfg, yaml.safe_load(f))
    _merge(cfg, overrides or {})
    return cfg


def build_scheme(args, env_info):
    """Scheme / groups / preprocess of run.py:97-119."""
    scheme = {
        "state": {"vshape": env_info["state_shape"]},
        "obs": {"vshape": env_info["obs_shape"], "group": "agents"},
        "actions": {"vshape": (1,), "group": "agents", "dtype": th.long},
        "avail_actions": {"vshape": (env_info["n_actions"],), "group": "agents", "dtype": th.int},
        "reward": {"vshape": (1,) if not args.ind_reward else (args.n_agents,)},
        "terminated": {"vshape": (1,), "dtype": th.uint8},
        "clean_num": {"vshape": (args.n_agents,)},
        "apple_den": {"vshape": (args.n_agents,)},
        "agent_pos": {"vshape": (args.n_agents, 2)},
        "agent_orientation": {"vshape": (args.n_agents, 2)},
    }
    if getattr(args, "obs_storage", "f32") == "code":      # compact storage: one u8 class code per window cell (include/ssd_hip.h)
        # (the env rejects the code format under extra_args.obs_color: "full")
        scheme["obs"] = {"vshape": tuple(env_info["obs_dims"]), "group": "agents", "dtype": th.uint8}
    if not getattr(args, "store_state", True):
        del scheme["state"]        # nothing downstream reads it; 22 MB/step at 4096 envs
    if "homophily" in args.name:
        scheme["actions_inc"] = {"vshape": (args.n_agents, 1), "group": "agents", "dtype": th.long}
    groups = {"agents": args.n_agents}
    preprocess = {"actions": ("actions_onehot", [OneHot(out_dim=args.n_actions)])}
    return scheme, groups, preprocess


def setup(config, logger=None):
    """Build runner / buffer / mac / learner exactly in the order of run.py:84-135."""
    args = SimpleNamespace(**config)
    if args.use_cuda and not th.cuda.is_available():
        args.use_cuda = False
    args.device = ("cuda:%d" % getattr(args, "device_index", 0)) if args.use_cuda else "cpu"
    # Units of the two schedules that the reference counts in env steps / episodes of its ONE env (epsilon_anneal_time,
    # target_update_interval).  "env_steps": the reference's literal arithmetic (default at batch_size_run == 1).  "rollouts"
    # (default for vectorised rollouts): one rollout of all envs advances the epsilon clock by episode_limit and the target-sync
    # counter counts learner.train calls -- the same number of rollouts / optimisation steps per anneal and per sync as the reference.
    if getattr(args, "schedule_unit", None) not in ("env_steps", "rollouts"):
        args.schedule_unit = "env_steps" if args.batch_size_run == 1 else "rollouts"
    args.train_steps_per_rollout = max(1, int(getattr(args, "train_steps_per_rollout", 1) or 1))
    if getattr(args, "strict_device_ops", False):
        from . import ops
        ops.set_strict(True)
    logger = logger or Logger()
    runner = r_REGISTRY[args.runner](args=args, logger=logger)
    env_info = runner.get_env_info()
    args.n_agents, args.n_actions = env_info["n_agents"], env_info["n_actions"]
    args.state_shape, args.obs_shape = env_info["state_shape"], env_info["obs_shape"]
    if args.rgb_input:
        args.state_dims, args.obs_dims = env_info["state_dims"], env_info["obs_dims"]
    scheme, groups, preprocess = build_scheme(args, env_info)
    buffer = ReplayBuffer(scheme, groups, args.buffer_size, env_info["episode_limit"] + 1, preprocess=preprocess,
                          device="cpu" if args.buffer_cpu_only else args.device)
    mac = mac_REGISTRY[args.mac](buffer.scheme, groups, args)
    runner.setup(scheme=scheme, groups=groups, preprocess=preprocess, mac=mac)
    learner = le_REGISTRY[args.learner](mac, buffer.scheme, logger, args)
    if args.use_cuda:
        learner.cuda()
    if args.schedule_unit == "rollouts":       # the log intervals run on the schedule clock too (see HomophilyLearner.log_clock)
        learner.log_clock = lambda: runner.sched_t
    if getattr(args, "replay_in_place", True) and hasattr(runner, "set_replay_buffer"):
        runner.set_replay_buffer(buffer)        # hip_graph: rollouts land in the replay buffer's own slots when the sizes allow
    return SimpleNamespace(args=args, logger=logger, runner=runner, buffer=buffer, mac=mac, learner=learner, train_steps=0)


def settle_gc():
    """Call once the loop's long-lived objects exist (networks, buffers, captured graphs): collect what setup left behind and move
    every surviving object into the permanent generation (gc.freeze).  A full collection of CPython's cyclic GC walks every tracked
    object -- with the tensors, modules and ctypes tables of this loop that is a 40-70 ms host pause, and at 8 train steps per
    rollout one of them lands every few dozen iterations; the host runs only ~5 iterations ahead of the GPU, so the device ran dry
    (bench trace, round 4: one train step of 55 ms on the device timeline).  Frozen objects are not walked again; the loop's own
    short-lived garbage stays cheap to collect."""
    import gc
    gc.collect()
    gc.freeze()


def train_iteration(ctx, episode):
    """One pass of the while-loop body of run.py:181-210: rollout, insert, sample, train (train_steps_per_rollout times; the
    reference's cadence is one).  The learner's episode counter (target sync every target_update_interval, homophily_learner.py:255-257)
    is the env-episode count under schedule_unit "env_steps" and the number of learner.train calls under "rollouts"."""
    a = ctx.args
    marks = getattr(ctx, "_trace_marks", None)          # diagnostic (bench SSD_BENCH_TRACE): device events between the phases

    def mark(tag):
        if marks is not None:
            e = th.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
    mark("start")
    batch = ctx.runner.run(test_mode=False)
    mark("rollout")
    ctx.buffer.insert_episode_batch(batch)
    if ctx.buffer.can_sample(a.batch_size):
        for _ in range(a.train_steps_per_rollout):
            sample = ctx.buffer.sample(a.batch_size, out=ctx.learner.sample_out() if hasattr(ctx.learner, "sample_out") else None)
            # run.py:188-189 trims the sample to its longest episode; the vectorised runners always write episode_limit + 1 slots
            # (`filled` is all ones), so the trim is the identity there and the device -> host read of max_t_filled() is skipped
            if not getattr(ctx.runner, "fixed_length_episodes", False):
                sample = sample[:, :sample.max_t_filled()]
            if str(sample.device) != str(a.device):
                sample.to(a.device)
            ctx.learner.train(sample, ctx.runner.t_env, ctx.train_steps if a.schedule_unit == "rollouts" else episode)
            ctx.train_steps += 1
            mark("train")
    return episode + a.batch_size_run


def run_sequential(config, logger=None):
    ctx = setup(config, logger)
    a, runner, learner, log = ctx.args, ctx.runner, ctx.learner, ctx.logger
    if a.checkpoint_path:
        steps = [int(d) for d in os.listdir(a.checkpoint_path) if d.isdigit() and os.path.isdir(os.path.join(a.checkpoint_path, d))]
        pick = max(steps) if a.load_step == 0 else min(steps, key=lambda x: abs(x - a.load_step))     # run.py:137-164
        learner.load_models(os.path.join(a.checkpoint_path, str(pick)))
        runner.t_env = pick
        if a.evaluate:
            for _ in range(a.test_nepisode):
                runner.run(test_mode=True)
            runner.close_env()
            return ctx
    episode, last_test_T, last_log_T, saved_at = 0, -a.test_interval - 1, 0, 0
    settled = 0
    while runner.t_env <= a.t_max:
        episode = train_iteration(ctx, episode)
        if settled < 2 and ctx.train_steps >= (4, 64)[settled]:      # after the graphs are captured, and once more when everything is warm
            settle_gc()
            settled += 1
        if (runner.t_env - last_test_T) / a.test_interval >= 1.0:
            last_test_T = runner.t_env
            for _ in range(max(1, a.test_nepisode // runner.batch_size)):
                runner.run(test_mode=True)
        if a.save_model and (runner.t_env - saved_at >= a.save_model_interval or saved_at == 0):
            saved_at = runner.t_env
            path = os.path.join(a.local_results_path, "models", getattr(a, "unique_token", a.name), str(runner.t_env))
            os.makedirs(path, exist_ok=True)
            learner.save_models(path)                                        # agent.th, opt_env.th, opt_inc.th
        if runner.t_env - last_log_T >= a.log_interval:
            log.log_stat("episode", episode, runner.t_env)
            log.print_recent_stats()
            last_log_T = runner.t_env
    runner.close_env()
    return ctx
