"""ctypes wrapper around oracle/libssd_oracle.so (TEST INFRASTRUCTURE -- never imported by the product package).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from homophily_marl_amd import abi
from homophily_marl_amd.envs.config import make_config

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_DIR, "libssd_oracle.so")
    src = os.path.join(_DIR, "ssd_oracle.c")
    hdr = os.path.join(_DIR, "..", "include", "ssd_hip.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _DIR, "-B", "libssd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_DIR, "libssd_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = abi.bind(C.CDLL(so), abi.CPU_SIGNATURES)
        _LIB.ssd_cpu_philox.restype = C.c_uint32
        _LIB.ssd_cpu_philox.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int]
        _LIB.ssd_cpu_counter_u32.restype = C.c_uint32
        _LIB.ssd_cpu_counter_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data


def make_tape(n_env, n_agents, max_uniforms, n_waste, move_order=None, uniforms=None, waste_order=None, spawn_rot=None,
              spawn_order=None):
    """Allocate (or wrap) tape arrays; returns (SsdTape, dict of arrays kept alive)."""
    arrs = dict(
        move_order=np.full((n_env, n_agents), 0xFF, np.uint8) if move_order is None else np.ascontiguousarray(move_order, np.uint8),
        uniforms=np.zeros((n_env, max(1, max_uniforms)), np.float64) if uniforms is None else np.ascontiguousarray(uniforms, np.float64),
        waste_order=np.zeros((n_env, max(1, n_waste)), np.uint8) if waste_order is None else np.ascontiguousarray(waste_order, np.uint8),
        spawn_rot=np.zeros((n_env, n_agents), np.uint8) if spawn_rot is None else np.ascontiguousarray(spawn_rot, np.uint8),
    )
    t = abi.SsdTape()
    t.move_order = _p(arrs["move_order"]); t.uniforms = _p(arrs["uniforms"])
    t.uniforms_stride = arrs["uniforms"].shape[1]
    t.waste_order = _p(arrs["waste_order"]); t.spawn_rot = _p(arrs["spawn_rot"])
    if spawn_order is not None:
        arrs["spawn_order"] = np.ascontiguousarray(spawn_order, np.uint8)      # [n_env, n_agents, n_spawn_points]
        t.spawn_order = _p(arrs["spawn_order"])
    t._keep = arrs
    return t, arrs


class OracleEnv:
    """Batch of envs stepped by the C restatement; numpy in/out."""

    def __init__(self, env, **kw):
        self.cfg, self.spec = make_config(env, **kw)
        self.lib = lib()
        h = C.c_void_p()
        abi.check(self.lib, self.lib.ssd_cpu_create(C.byref(self.cfg), C.byref(h)), "ssd_cpu_last_error")
        self.h = h
        info = abi.SsdInfo()
        abi.check(self.lib, self.lib.ssd_cpu_get_info(self.h, C.byref(info)), "ssd_cpu_last_error")
        self.info = info
        self.n_env, self.n = self.cfg.n_env, self.cfg.n_agents
        self.H, self.W, self.V = self.cfg.height, self.cfg.width, info.obs_edge
        self.n_actions = info.n_actions

    def close(self):
        if getattr(self, "h", None):          # the constructor may have failed before the handle existed
            self.lib.ssd_cpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out(self):
        N, n = self.n_env, self.n
        o = dict(reward=np.zeros((N, n), np.float32), clean_num=np.zeros((N, n), np.float32),
                 apple_den=np.zeros((N, n), np.float32), terminated=np.zeros(N, np.uint8),
                 collective_return=np.zeros(N, np.float32), equality=np.zeros(N, np.float32),
                 n_draws=np.zeros(N, np.int32))
        s = abi.SsdStepOut()
        for k, v in o.items():
            setattr(s, k, _p(v))
        return s, o

    def reset(self, tape=None, env_mask=None):
        s, o = self._out()
        m = None if env_mask is None else np.ascontiguousarray(env_mask, np.uint8)
        abi.check(self.lib, self.lib.ssd_cpu_reset(self.h, _p(m), C.byref(tape) if tape is not None else None, C.byref(s)),
                  "ssd_cpu_last_error")
        return o

    def step(self, actions, tape=None):
        a = np.ascontiguousarray(actions, np.int32).reshape(self.n_env, self.n)
        s, o = self._out()
        abi.check(self.lib, self.lib.ssd_cpu_step(self.h, _p(a), C.byref(tape) if tape is not None else None, C.byref(s)),
                  "ssd_cpu_last_error")
        return o

    def observe(self, fmt=abi.OBS_F32, want_state=False):
        N, n, V = self.n_env, self.n, self.V
        if fmt == abi.OBS_CODE:
            obs = np.zeros((N, n, V, V), np.uint8)
        else:
            obs = np.zeros((N, n, 3, V, V), {abi.OBS_F32: np.float32, abi.OBS_BF16: np.uint16, abi.OBS_U8: np.uint8}[fmt])
        o = dict(obs=obs, pos=np.zeros((N, n, 2), np.float32), orient=np.zeros((N, n, 2), np.float32))
        if want_state:
            o["state"] = np.zeros((N, 3, self.H, self.W), np.float32)
        s = abi.SsdObsOut()
        s.obs = _p(obs); s.obs_format = fmt; s.pos = _p(o["pos"]); s.orient = _p(o["orient"])
        s.state = _p(o.get("state"))
        abi.check(self.lib, self.lib.ssd_cpu_observe(self.h, C.byref(s)), "ssd_cpu_last_error")
        return o

    def export_state(self):
        N, n = self.n_env, self.n
        d = dict(grid=np.zeros((N, self.H * self.W), np.uint8), pos=np.zeros((N, n, 2), np.int16),
                 orient=np.zeros((N, n), np.uint8), ep_reward=np.zeros((N, n), np.int32),
                 ep_step=np.zeros(N, np.int32), epoch=np.zeros(N, np.uint32))
        s = abi.SsdState()
        for k, v in d.items():
            setattr(s, k, _p(v))
        abi.check(self.lib, self.lib.ssd_cpu_export_state(self.h, C.byref(s)), "ssd_cpu_last_error")
        return d

    def import_state(self, **arrays):
        s = abi.SsdState()
        keep = []
        dt = dict(grid=np.uint8, pos=np.int16, orient=np.uint8, ep_reward=np.int32, ep_step=np.int32, epoch=np.uint32)
        for k, v in arrays.items():
            a = np.ascontiguousarray(v, dt[k]); keep.append(a)
            setattr(s, k, _p(a))
        abi.check(self.lib, self.lib.ssd_cpu_import_state(self.h, C.byref(s)), "ssd_cpu_last_error")


def philox(c0, c1, c2, c3, seed, word=0):
    return lib().ssd_cpu_philox(c0, c1, c2, c3, seed, word)


def counter_u32(seed, env, episode, c, stream, k):
    return lib().ssd_cpu_counter_u32(seed, env, episode, c, stream, k)
