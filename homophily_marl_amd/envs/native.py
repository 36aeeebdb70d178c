"""NativeEnv: a batch of SSD envs resident on one MI355X, driven through the C ABI of libssd_hip.so.

PyTorch is plumbing here: it owns the device buffers that are handed to the library as raw pointers and supplies the
HIP stream.  All dynamics run in the hand-written kernels of csrc/ssd_env.hip; there is no CPU fallback.
"""
import ctypes as C

import torch

from .. import abi
from .config import make_config

_OBS_DTYPE = {abi.OBS_F32: torch.float32, abi.OBS_BF16: torch.bfloat16, abi.OBS_U8: torch.uint8, abi.OBS_CODE: torch.uint8}


def _ptr(t):
    return None if t is None else t.data_ptr()


class NativeEnv:
    def __init__(self, env, device=0, **kw):
        if not torch.cuda.is_available():
            raise RuntimeError("NativeEnv needs a HIP device (no CPU fallback on the product path)")
        self.lib = abi.load_library()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.cfg, self.spec = make_config(env, device=self.device.index, **kw)
        h = C.c_void_p()
        abi.check(self.lib, self.lib.ssd_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        info = abi.SsdInfo()
        abi.check(self.lib, self.lib.ssd_get_info(self.h, C.byref(info)))
        self.info = info
        self.n_env, self.n = self.cfg.n_env, self.cfg.n_agents
        self.H, self.W, self.V = self.cfg.height, self.cfg.width, info.obs_edge
        self.n_actions = info.n_actions
        self.tape_mode = self.cfg.rng_mode == abi.RNG_TAPE
        N, n = self.n_env, self.n
        f32 = dict(dtype=torch.float32, device=self.device)
        # persistent output buffers (the library writes into caller-owned memory)
        self.out = dict(reward=torch.zeros(N, n, **f32), clean_num=torch.zeros(N, n, **f32),
                        apple_den=torch.zeros(N, n, **f32), terminated=torch.zeros(N, dtype=torch.uint8, device=self.device),
                        collective_return=torch.zeros(N, **f32), equality=torch.zeros(N, **f32),
                        n_draws=torch.zeros(N, dtype=torch.int32, device=self.device))
        self._so = abi.SsdStepOut()
        for k, v in self.out.items():
            setattr(self._so, k, v.data_ptr())
        self._obs_bufs = {}

    # ------------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize(self.device)
            self.lib.ssd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _tape(self, tape):
        if tape is None:
            if self.tape_mode:
                raise abi.SsdError("TAPE mode needs a tape")
            return None
        return C.byref(tape)

    def make_tape(self, move_order=None, uniforms=None, waste_order=None, spawn_rot=None, spawn_order=None):
        """Wrap device (or host -> copied) arrays into an ssd_tape; returns the struct (keeps the tensors alive)."""
        def dev(x, dt):
            if x is None:
                return None
            return torch.as_tensor(x).to(device=self.device, dtype=dt).contiguous()
        t = abi.SsdTape()
        keep = dict(move_order=dev(move_order, torch.uint8), uniforms=dev(uniforms, torch.float64),
                    waste_order=dev(waste_order, torch.uint8), spawn_rot=dev(spawn_rot, torch.uint8),
                    spawn_order=dev(spawn_order, torch.uint8))
        t.move_order, t.uniforms = _ptr(keep["move_order"]), _ptr(keep["uniforms"])
        t.uniforms_stride = keep["uniforms"].shape[1] if keep["uniforms"] is not None else 0
        t.waste_order, t.spawn_rot = _ptr(keep["waste_order"]), _ptr(keep["spawn_rot"])
        t.spawn_order = _ptr(keep["spawn_order"])
        t._keep = keep
        return t

    def obs_buffers(self, fmt=abi.OBS_F32, want_state=False, want_code=False):
        """Persistent output buffers of an observation pass.  want_code (simplified colours, fmt != OBS_CODE): also the window as u8
        class codes [n_env, n, V * V rounded up to 16] (ssd_obs_out.obs_code: what the rollout-time encoder reads)."""
        key = (fmt, want_state, want_code)
        if key not in self._obs_bufs:
            N, n, V = self.n_env, self.n, self.V
            shape = (N, n, V, V) if fmt == abi.OBS_CODE else (N, n, 3, V, V)
            b = dict(obs=torch.empty(shape, dtype=_OBS_DTYPE[fmt], device=self.device),
                     pos=torch.empty(N, n, 2, dtype=torch.float32, device=self.device),
                     orient=torch.empty(N, n, 2, dtype=torch.float32, device=self.device))
            if want_state:
                b["state"] = torch.empty(N, 3, self.H, self.W, dtype=torch.float32, device=self.device)
            if want_code:
                assert fmt != abi.OBS_CODE
                b["code"] = torch.zeros(N, n, abi.code_agent_stride(V), dtype=torch.uint8, device=self.device)
            self._obs_bufs[key] = b
        return self._obs_bufs[key]

    def _oo(self, bufs, fmt):
        o = abi.SsdObsOut()
        o.obs, o.obs_format = _ptr(bufs.get("obs")), fmt
        o.state, o.pos, o.orient = _ptr(bufs.get("state")), _ptr(bufs.get("pos")), _ptr(bufs.get("orient"))
        # obs placed inside an episode storage [n_env, t_slots, n, ...]: the env writes slot ep_step itself
        o.obs_env_stride, o.obs_slot_stride = int(bufs.get("obs_env_stride", 0)), int(bufs.get("obs_slot_stride", 0))
        o.obs_t_slots = int(bufs.get("obs_t_slots", 0))
        o.obs_code = _ptr(bufs.get("code"))
        return o

    def storage_obs_buffers(self, storage, fmt=abi.OBS_F32, want_code=False):
        """Output buffers whose `obs` is an episode storage [n_env, t_slots, n, 3, V, V] (dense inner dims): every observe /
        step_observe writes the env's block at time slot ep_step (include/ssd_hip.h: ssd_obs_out.obs_env_stride); a call that would
        land past the last slot writes nothing and raises error bit 16 (obs_t_slots).  want_code: plus the dense class-code side
        buffer of obs_buffers."""
        N, n, V = self.n_env, self.n, self.V
        assert storage.shape[0] == N and tuple(storage.shape[2:]) == ((n, V, V) if fmt == abi.OBS_CODE else (n, 3, V, V))
        assert storage[0, 0].is_contiguous() and storage.dtype == _OBS_DTYPE[fmt] and storage.device == self.device
        dense = self.obs_buffers(fmt, want_code=want_code)
        b = dict(obs=storage, pos=dense["pos"], orient=dense["orient"], obs_env_stride=storage.stride(0), obs_slot_stride=storage.stride(1),
                 obs_t_slots=storage.shape[1])
        if want_code:
            b["code"] = dense["code"]
        return b

    # ------------------------------------------------------------------------------------------------------
    def reset(self, tape=None, env_mask=None):
        m = None if env_mask is None else torch.as_tensor(env_mask).to(device=self.device, dtype=torch.uint8).contiguous()
        abi.check(self.lib, self.lib.ssd_reset(self.h, _ptr(m), self._tape(tape), C.byref(self._so), self._stream()))
        return self.out

    def step(self, actions, tape=None):
        a = self._actions(actions)
        abi.check(self.lib, self.lib.ssd_step(self.h, a.data_ptr(), self._tape(tape), C.byref(self._so), self._stream()))
        return self.out

    def observe(self, fmt=abi.OBS_F32, want_state=False, out=None):
        bufs = out if out is not None else self.obs_buffers(fmt, want_state)
        oo = self._oo(bufs, fmt)
        abi.check(self.lib, self.lib.ssd_observe(self.h, C.byref(oo), self._stream()))
        return bufs

    def step_observe(self, actions, tape=None, fmt=abi.OBS_F32, out=None):
        """env.step followed by the next get_obs, fused in one launch.  Returns the step outputs merged with the
        observation buffers of the NEW state."""
        a = self._actions(actions)
        bufs = out if out is not None else self.obs_buffers(fmt, False)
        oo = self._oo(bufs, fmt)
        abi.check(self.lib, self.lib.ssd_step_observe(self.h, a.data_ptr(), self._tape(tape), C.byref(self._so), C.byref(oo),
                                                      self._stream()))
        r = dict(self.out)
        r.update(bufs)
        return r

    def _actions(self, actions):
        a = actions
        if not isinstance(a, torch.Tensor) or a.device != self.device or a.dtype != torch.int32 or not a.is_contiguous():
            a = torch.as_tensor(actions).to(device=self.device, dtype=torch.int32).contiguous()
        if a.numel() != self.n_env * self.n:
            raise abi.SsdError("actions must have n_env * n_agents elements")
        self._last_actions = a  # keep alive until the launch has consumed it
        return a

    def poll_error(self):
        bits = C.c_int32(0)
        abi.check(self.lib, self.lib.ssd_poll_error(self.h, C.byref(bits)))
        return bits.value

    def export_state(self):
        N, n = self.n_env, self.n
        d = dict(grid=torch.empty(N, self.H * self.W, dtype=torch.uint8, device=self.device),
                 pos=torch.empty(N, n, 2, dtype=torch.int16, device=self.device),
                 orient=torch.empty(N, n, dtype=torch.uint8, device=self.device),
                 ep_reward=torch.empty(N, n, dtype=torch.int32, device=self.device),
                 ep_step=torch.empty(N, dtype=torch.int32, device=self.device),
                 epoch=torch.empty(N, dtype=torch.int32, device=self.device))
        s = abi.SsdState()
        for k, v in d.items():
            setattr(s, k, v.data_ptr())
        abi.check(self.lib, self.lib.ssd_export_state(self.h, C.byref(s), self._stream()))
        return d

    def import_state(self, **arrays):
        dt = dict(grid=torch.uint8, pos=torch.int16, orient=torch.uint8, ep_reward=torch.int32, ep_step=torch.int32,
                  epoch=torch.int32)
        s = abi.SsdState()
        keep = []
        for k, v in arrays.items():
            t = torch.as_tensor(v).to(device=self.device, dtype=dt[k]).contiguous()
            keep.append(t)
            setattr(s, k, t.data_ptr())
        abi.check(self.lib, self.lib.ssd_import_state(self.h, C.byref(s), self._stream()))
        self._keep_import = keep
