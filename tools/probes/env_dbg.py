"""Debug probe: first reset of a golden trajectory on the HIP env, printed next to the fixture."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from tests import golden_util as GU
from tests.hip_adapter import HipEnv, hip_tape
from homophily_marl_amd import abi

path = [p for p in GU.traj_files() if "cleanup10_allact" in p][0]
z, meta = GU.load(path)
n = meta["num_agents"]
env = HipEnv(meta["env"], **GU.env_kwargs(meta, 1))
U = z["uniforms"].shape[1]; Wn = z["waste_order"].shape[1]
rep = lambda a: np.repeat(np.asarray(a)[None], 1, 0)
c = 0
tape = hip_tape(1, n, U, Wn, rep(z["move_order"][c]), rep(z["uniforms"][c]), rep(z["waste_order"][c]), rep(z["spawn_rot"][c]),
                *((rep(z["spawn_order"][c]),) if "spawn_order" in z else ()))
o = env.reset(tape)
st = env.export_state()
print("n_draws", o["n_draws"], "expected", z["n_uniforms"][c], "err", env.e.poll_error())
print("grid equal", (st["grid"][0] == z["grid"][c].reshape(-1)).mean(), "pos", st["pos"][0].tolist(), z["pos"][c].tolist())
print("grid hist", np.bincount(st["grid"][0], minlength=8), np.bincount(z["grid"][c].reshape(-1), minlength=8))
print("epoch", st["epoch"], "ep_step", st["ep_step"])
env2 = HipEnv(meta["env"], **GU.env_kwargs(meta, 4, rng_mode=abi.RNG_COUNTER, seed=3))
o2 = env2.reset()
print("counter n_draws", o2["n_draws"], "err", env2.e.poll_error())
