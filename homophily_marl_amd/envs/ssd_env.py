"""Cleanup / Harvest behind the reference's env plugin surface (src/envs/__init__.py:6-11: REGISTRY[name](**env_args)).

`CleanupHipEnv(**env_args)` / `HarvestHipEnv(**env_args)` take the reference constructor's kwargs (cleanup.py:29,
harvest.py:18) and expose the MultiAgentEnv methods plus get_agent_pos / get_agent_orientation (map_env.py:917-921).
With the default `n_env=1` every method returns what the reference returns (numpy, one env), so the class drops into
the reference's EpisodeRunner.  With `n_env > 1` the same object is a batch of envs and the `*_batch` methods return
device tensors without any host round trip (used by HipVecRunner).  All dynamics run in libssd_hip.so.
"""
import numpy as np
import torch

from .. import abi
from .multiagentenv import MultiAgentEnv
from .native import NativeEnv


class SSDHipEnv(MultiAgentEnv):
    ENV = None

    def __init__(self, ascii_map=None, num_agents=1, render=False, seed=None, episode_limit=100, is_replay=False, view_size=7,
                 map="default", extra_args=None, n_env=1, device=0, rng_mode=abi.RNG_COUNTER, env_id_base=0):
        if render or is_replay:
            raise NotImplementedError("rendering / replays are out of scope (SURVEY.md section 2, row 21)")
        self.native = NativeEnv(self.ENV, device=device, map=map, num_agents=num_agents, n_env=n_env, view_size=view_size,
                                episode_limit=episode_limit, extra_args=extra_args, rng_mode=rng_mode,
                                seed=0 if seed is None else seed, env_id_base=env_id_base)
        self.extra_args = dict(self.native_extra_args(extra_args))
        self.n_env = n_env
        self.num_agents = self.n_agents = num_agents
        self.n_actions = self.native.n_actions
        self.episode_limit = episode_limit
        self.view_size = view_size
        self.env_name = self.ENV
        self.device = self.native.device
        n = self.n_agents
        avail = [1] * self.n_actions                                   # get_avail_agent_actions (map_env.py:972-980)
        if self.extra_args["disable_rotation_action"]:
            avail[5] = avail[6] = 0
        if self.extra_args["disable_fire_action"]:
            avail[7] = 0
        self._avail = avail
        self.avail_actions_batch = torch.tensor(avail, dtype=torch.int32, device=self.device).expand(n_env, n, -1).contiguous()
        self._tape = None
        self._last = None

    @staticmethod
    def native_extra_args(extra_args):
        from .config import DEFAULT_EXTRA_ARGS
        ea = dict(DEFAULT_EXTRA_ARGS)
        ea.update(extra_args or {})
        return ea

    def set_tape(self, tape):
        """TAPE mode: the recorded draws the next reset()/step() consumes."""
        self._tape = tape

    # ---- batch API (device tensors) -----------------------------------------------------------------------------
    def reset_batch(self, env_mask=None):
        self.native.reset(self._tape, env_mask)

    def step_batch(self, actions, observe=True, fmt=abi.OBS_F32, out=None):
        """actions int32 [n_env, n] on the device.  Returns dict(reward, clean_num, apple_den [n_env, n] f32,
        terminated u8 [n_env], collective_return, equality f32 [n_env]) and, with observe=True, the observation of the
        NEW state (obs, pos, orient) from the same launch."""
        if observe:
            self._last = self.native.step_observe(actions, self._tape, fmt, out=out)   # out: e.g. native.storage_obs_buffers(...)
        else:
            self._last = self.native.step(actions, self._tape)
        return self._last

    def observe_batch(self, fmt=abi.OBS_F32, want_state=False, out=None):
        return self.native.observe(fmt, want_state, out=out)

    # ---- reference single-env API -------------------------------------------------------------------------------
    def _one(self):
        if self.n_env != 1:
            raise RuntimeError("the reference-shaped methods need n_env == 1; use the *_batch methods")

    def reset(self):
        self.native.reset(self._tape)

    def step(self, actions):
        """Returns reward f64[n], terminated bool, info (map_env.py:874-915)."""
        self._one()
        a = torch.as_tensor(np.asarray([int(x) for x in actions], dtype=np.int32)).reshape(1, self.n_agents)
        o = self.native.step(a, self._tape)
        bits = self.native.poll_error()
        if bits & 1:
            raise KeyError("action out of range (reference: KeyError in action_map, agent.py:174-176,235-237)")
        reward = o["reward"][0].double().cpu().numpy()
        terminated = bool(o["terminated"][0].item())
        info = {}
        if terminated:
            info["collective_return"] = float(o["collective_return"][0].item())
            info["equality_metric"] = float(o["equality"][0].item())
        info["clean_num"] = o["clean_num"][0].double().cpu().numpy()
        info["apple_den"] = o["apple_den"][0].double().cpu().numpy()
        return reward, terminated, info

    def get_obs(self):
        self._one()
        o = self.native.observe(abi.OBS_F32)["obs"][0].double().cpu().numpy()
        return [o[i] for i in range(self.n_agents)]

    def get_obs_agent(self, agent_id):
        return self.get_obs()[agent_id]

    def get_obs_size(self):
        return (3, self.native.V, self.native.V)

    def get_state(self):
        self._one()
        return self.native.observe(abi.OBS_F32, want_state=True)["state"][0].double().cpu().numpy()

    def get_state_size(self):
        return (3, self.native.H, self.native.W)

    def get_avail_actions(self):
        return [list(self._avail) for _ in range(self.n_agents)]

    def get_avail_agent_actions(self, agent_id):
        return list(self._avail)

    def get_total_actions(self):
        return self.n_actions

    def get_agent_pos(self):
        self._one()
        return self.native.observe(abi.OBS_F32, out=self._pos_only())["pos"][0].double().cpu().numpy()

    def get_agent_orientation(self):
        self._one()
        return self.native.observe(abi.OBS_F32, out=self._pos_only())["orient"][0].double().cpu().numpy()

    def _pos_only(self):
        b = self.native.obs_buffers(abi.OBS_F32)
        return dict(pos=b["pos"], orient=b["orient"])

    def get_env_info(self):
        info = MultiAgentEnv.get_env_info(self)
        info["state_dims"] = (self.native.H, self.native.W)                   # map_env.py:1016-1017
        info["obs_dims"] = (self.native.V, self.native.V)
        return info

    def get_stats(self):
        return {}

    def render(self):
        raise NotImplementedError("rendering is out of scope")

    def save_replay(self):
        raise NotImplementedError("replays are out of scope")

    def seed(self):
        return None

    def close(self):
        self.native.close()


class CleanupHipEnv(SSDHipEnv):
    ENV = "cleanup"


class HarvestHipEnv(SSDHipEnv):
    ENV = "harvest"
