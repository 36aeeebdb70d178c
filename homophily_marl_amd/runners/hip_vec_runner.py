"""HipVecRunner: episode rollouts of N vectorised envs resident on the GPU, behind the reference's runner surface
(src/runners/episode_runner.py:7-152: __init__(args, logger), setup, get_env_info, run(test_mode) -> EpisodeBatch,
reset, close_env, save_replay, attrs t_env / batch_size).

Per timestep (episode_runner.py:57-97): store obs -> env-head action selection -> env.step -> store rewards ->
incentive-head action selection -> store.  Here the env transition and the observation of the NEXT state come from one
fused kernel launch (ssd_step_observe) and nothing leaves the device.  `runner: "episode"` is the same loop with the
reference's restriction batch_size_run == 1.
"""
from functools import partial

import torch as th

from .. import abi
from ..components.episode_buffer import EpisodeBatch
from ..envs import REGISTRY as env_REGISTRY


class HipVecRunner:
    single_env_only = False
    fixed_length_episodes = True       # the SSD envs terminate at episode_limit only: every stored episode fills all T + 1 slots

    def __init__(self, args, logger):
        self.args, self.logger = args, logger
        self.batch_size = args.batch_size_run
        if self.single_env_only:
            assert self.batch_size == 1                                   # episode_runner.py:13
        env_args = dict(args.env_args)
        env_args.setdefault("n_env", self.batch_size)
        env_args.setdefault("device", getattr(args, "device_index", 0))
        env_args.setdefault("env_id_base", getattr(args, "env_id_base", 0))
        self.env = env_REGISTRY[args.env](**env_args)
        self.episode_limit = self.env.episode_limit
        self.t = 0
        self.t_env = 0
        self.rollouts = 0          # training rollouts finished (schedule_unit "rollouts": the epsilon clock, see sched_t)
        self.train_returns, self.test_returns = [], []          # reference attributes; the statistics live on the device (_finish_stats)
        self.train_stats, self.test_stats = {}, {}
        self.log_train_stats_t = -1000000
        # obs_storage: "code" keeps observations as u8 class codes (simplified palette; 12x fewer bytes in the storage and the
        # replay buffer); the controller expands them where it consumes them
        self.obs_fmt = abi.OBS_CODE if getattr(self.args, "obs_storage", "f32") == "code" else abi.OBS_F32

    @property
    def sched_t(self):
        """The clock of the epsilon schedule (epsilon_schedules.py; reference: t_env, episode_runner.py:72).  The reference only
        ever runs ONE env, where t_env advances by episode_limit per rollout and per learner.train.  schedule_unit "env_steps"
        keeps that literally; "rollouts" (the default for batch_size_run > 1, run.py setup) advances the clock by episode_limit per
        ROLLOUT, so that epsilon_anneal_time spans the same number of rollouts / optimisation steps as in the reference instead of
        collapsing to the first rollout (4096 envs x 100 steps >> 50000)."""
        if getattr(self.args, "schedule_unit", "env_steps") == "rollouts":
            return self.rollouts * self.episode_limit
        return self.t_env

    def setup(self, scheme, groups, preprocess, mac):
        self.new_batch = partial(EpisodeBatch, scheme, groups, self.batch_size, self.episode_limit + 1, preprocess=preprocess,
                                 device=self.args.device)
        self.mac = mac
        self.store_state = "state" in scheme and getattr(self.args, "store_state", True)

    def get_env_info(self):
        return self.env.get_env_info()

    def save_replay(self):
        self.env.save_replay()

    def close_env(self):
        self.env.close()

    def reset(self):
        self.batch = self.new_batch()
        self.env.reset_batch()
        self.t = 0

    def _store_observation(self, o, t):
        data = {"avail_actions": self.env.avail_actions_batch, "obs": o["obs"], "agent_pos": o["pos"], "agent_orientation": o["orient"]}
        if self.store_state:
            data["state"] = self.env.observe_batch(self.obs_fmt, want_state=True)["state"]
        self.batch.update(data, ts=t)

    # The rollout is exposed timestep by timestep (begin_episode / step_once / finish_episode) so that a caller such as
    # bench.py can time an exact number of transitions; run() is the reference-shaped whole-episode call.
    def begin_episode(self, test_mode=False):
        self.reset()
        self._test_mode = test_mode
        self._ep_return = th.zeros(self.batch_size, self.args.n_agents, device=self.env.device)
        self.mac.init_hidden(batch_size=self.batch_size)
        self._o = self.env.observe_batch(self.obs_fmt)
        self._out = None

    @th.no_grad()
    def step_once(self):
        t, test_mode = self.t, self._test_mode
        self._store_observation(self._o, t)
        actions = self.mac.select_actions_env(self.batch, t_ep=t, t_env=self.sched_t, test_mode=test_mode)
        out = self.env.step_batch((actions.squeeze(-1) % self.args.n_actions).to(th.int32), observe=True, fmt=self.obs_fmt)
        self._ep_return += out["reward"]
        # `terminated` is stored as the env flag: episode_limit never appears in info (episode_runner.py:83)
        self.batch.update({"actions": actions, "reward": out["reward"], "terminated": out["terminated"].unsqueeze(-1),
                           "clean_num": out["clean_num"], "apple_den": out["apple_den"]}, ts=t)
        actions_inc = self.mac.select_actions_inc(actions, self.batch, t_ep=t, t_env=self.sched_t, test_mode=test_mode)
        self.batch.update({"actions_inc": actions_inc}, ts=t)
        self._o = self._out = out
        self.t += 1
        return self.t >= self.episode_limit

    @th.no_grad()
    def finish_episode(self):
        """slot T: last observation and the bootstrapping actions (episode_runner.py:99-119), then stats."""
        test_mode = self._test_mode
        self._store_observation(self._o, self.t)
        actions = self.mac.select_actions_env(self.batch, t_ep=self.t, t_env=self.sched_t, test_mode=test_mode)
        actions_inc = self.mac.select_actions_inc(actions, self.batch, t_ep=self.t, t_env=self.sched_t, test_mode=test_mode)
        self.batch.update({"actions_inc": actions_inc}, ts=self.t)
        self.batch.update({"actions": actions}, ts=self.t)
        return self._finish_stats()

    def _finish_stats(self):
        """Episode statistics of episode_runner.py:121-152 (collective_return / equality_metric / ep_length sums over the episodes,
        mean and standard deviation of the per-agent returns), accumulated ON THE DEVICE: a rollout adds its sums to a small f64
        tensor with a handful of device ops and no host synchronisation; the host reads the tensor once per log interval (_log)."""
        test_mode = self._test_mode
        out, ep_return = self._out, self._ep_return
        stats = self.test_stats if test_mode else self.train_stats
        prefix = "test_" if test_mode else ""
        if getattr(self.args, "runner_stats", True):
            if not getattr(self, "_stats_on_device_done", False):
                self._stats_device(out, ep_return, test_mode)
            self._stats_on_device_done = False
            stats["n_episodes"] = self.batch_size + stats.get("n_episodes", 0)
            stats["ep_length"] = self.t * self.batch_size + stats.get("ep_length", 0)
            stats["n_returns"] = ep_return.numel() + stats.get("n_returns", 0)
        if not test_mode:
            self.t_env += self.t * self.batch_size
            self.rollouts += 1
        if getattr(self.args, "runner_stats", True):
            if test_mode and stats["n_episodes"] >= self.args.test_nepisode:
                self._log(stats, prefix, test_mode)
            # runner_log_interval is measured on the schedule clock: under schedule_unit "rollouts" a rollout of all envs advances it
            # by episode_limit, so the host reads the device sums every runner_log_interval / episode_limit rollouts (the reference's
            # cadence) instead of after every rollout (4096 envs advance t_env by 409 600)
            elif not test_mode and self.sched_t - self.log_train_stats_t >= self.args.runner_log_interval:
                self._log(stats, prefix, test_mode)
                if hasattr(self.mac.action_selector, "epsilon"):
                    self.logger.log_stat("epsilon", self.mac.action_selector.epsilon, self.t_env)
                self.log_train_stats_t = self.sched_t
        return self.batch

    def _stats_device(self, out, ep_return, test_mode):
        """the device side of _finish_stats: this rollout's sums added to the accumulator (a handful of launches, no synchronisation;
        the graph runner replays them as part of its episode-closing graph)"""
        from .. import ops
        ops.runner_stats(out["collective_return"], out["equality"], ep_return, self._stat_acc(test_mode))

    def _stat_acc(self, test_mode):
        """device accumulator [sum collective_return, sum equality_metric, sum of returns, sum of squared returns]"""
        key = "_acc_test" if test_mode else "_acc_train"
        if getattr(self, key, None) is None:
            setattr(self, key, th.zeros(4, dtype=th.float64, device=self.env.device))
        return getattr(self, key)

    def run(self, test_mode=False):
        self.begin_episode(test_mode)
        while not self.step_once():
            pass
        return self.finish_episode()

    def _log(self, stats, prefix, test_mode):
        acc = self._stat_acc(test_mode)
        coll, eq, rs, rss = acc.cpu().tolist()          # the one device -> host read of the log interval
        acc.zero_()
        n_ep, n_ret = stats["n_episodes"], max(1, stats.get("n_returns", 0))
        mean = rs / n_ret
        self.logger.log_stat(prefix + "return_mean", mean, self.t_env)
        self.logger.log_stat(prefix + "return_std", max(0.0, rss / n_ret - mean * mean) ** 0.5, self.t_env)    # np.std: population
        for k, v in (("collective_return", coll), ("equality_metric", eq), ("ep_length", stats.get("ep_length", 0))):
            self.logger.log_stat(prefix + k + "_mean", v / n_ep, self.t_env)
        stats.clear()


class EpisodeRunner(HipVecRunner):
    """`runner: "episode"`: one env, as the reference asserts (episode_runner.py:13)."""
    single_env_only = True
