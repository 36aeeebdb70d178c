"""numpy-facing adapter around NativeEnv so the golden replayer and the oracle-comparison tests can drive the HIP
library exactly like oracle.oracle_py.OracleEnv."""
import numpy as np
import torch

from homophily_marl_amd import abi
from homophily_marl_amd.envs.native import NativeEnv


def _np(d):
    out = {}
    for k, v in d.items():
        a = v.detach().cpu()
        if a.dtype == torch.bfloat16:
            a = a.view(torch.int16).numpy().view(np.uint16)
        else:
            a = a.numpy()
        out[k] = a
    return out


def hip_tape(n_env, n, U, Wn, mo=None, un=None, wo=None, sr=None, so=None):
    dev = torch.device("cuda", 0)
    keep = dict(
        move_order=torch.as_tensor(np.full((n_env, n), 0xFF, np.uint8) if mo is None else np.ascontiguousarray(mo, np.uint8)).to(dev),
        uniforms=torch.as_tensor(np.zeros((n_env, max(1, U))) if un is None else np.ascontiguousarray(un, np.float64)).to(dev),
        waste_order=torch.as_tensor(np.zeros((n_env, max(1, Wn)), np.uint8) if wo is None else np.ascontiguousarray(wo, np.uint8)).to(dev),
        spawn_rot=torch.as_tensor(np.zeros((n_env, n), np.uint8) if sr is None else np.ascontiguousarray(sr, np.uint8)).to(dev))
    t = abi.SsdTape()
    t.move_order, t.uniforms = keep["move_order"].data_ptr(), keep["uniforms"].data_ptr()
    t.uniforms_stride = keep["uniforms"].shape[1]
    t.waste_order, t.spawn_rot = keep["waste_order"].data_ptr(), keep["spawn_rot"].data_ptr()
    if so is not None:
        keep["spawn_order"] = torch.as_tensor(np.ascontiguousarray(so, np.uint8)).to(dev)
        t.spawn_order = keep["spawn_order"].data_ptr()
    t._keep = keep
    return t


class HipEnv:
    def __init__(self, env, **kw):
        self.e = NativeEnv(env, device=0, **kw)
        self.info = self.e.info
        self.n_env, self.n, self.n_actions = self.e.n_env, self.e.n, self.e.n_actions
        self.H, self.W, self.V = self.e.H, self.e.W, self.e.V

    def close(self):
        bits = self.e.poll_error()
        self.e.close()
        assert bits == 0, "device error bits %d" % bits

    def reset(self, tape=None, env_mask=None):
        return _np(self.e.reset(tape, env_mask))

    def step(self, actions, tape=None):
        return _np(self.e.step(np.ascontiguousarray(actions, np.int32), tape))

    def step_observe(self, actions, tape=None, fmt=abi.OBS_F32):
        return _np(self.e.step_observe(np.ascontiguousarray(actions, np.int32), tape, fmt))

    def observe(self, fmt=abi.OBS_F32, want_state=False):
        return _np(self.e.observe(fmt, want_state))

    def export_state(self):
        d = _np(self.e.export_state())
        d["epoch"] = d["epoch"].view(np.uint32)
        return d

    def import_state(self, **arrays):
        if "epoch" in arrays:
            arrays["epoch"] = np.asarray(arrays["epoch"]).astype(np.uint32).view(np.int32)
        self.e.import_state(**{k: np.ascontiguousarray(v) for k, v in arrays.items()})
