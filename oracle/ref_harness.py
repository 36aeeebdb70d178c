"""Harness around the *imported reference* env (TEST INFRASTRUCTURE; runs only in the build container).

It never travels to the GPU box: /root/reference does not exist there.  Its only products are the committed
fixtures under tests/golden/ (written by oracle/gen_golden.py) and the pass/fail of oracle/check_vs_reference.py.

The reference draws randomness from three global sources:
  np.random.shuffle (map_env.py:541), np.random.rand (cleanup.py:172,183; harvest.py:119),
  np.random.randint (map_env.py:789) and Python's random.shuffle (cleanup.py:178; map_env.py:777).
`RefEnv` wraps exactly these four functions while a reset()/step() call is running:
  mode="tape"    -> call the original, record what it produced (the tape the oracle / HIP kernels replay);
  mode="counter" -> replace the draw by the COUNTER-mode generator of include/ssd_hip.h (logic untouched),
                    and record it as well.
"""
import contextlib
import io
import random
import sys
import types

import numpy as np

REF_SRC = "/root/reference/src"

STREAM_UNIFORM, STREAM_MOVE, STREAM_WASTE, STREAM_SPAWN_ROT = 0, 1, 2, 3
CODES = {" ": 0, "@": 1, "A": 2, "H": 3, "R": 4, "S": 5}
ORI = {"LEFT": 0, "RIGHT": 1, "UP": 2, "DOWN": 3}
ORI_NAMES = ["LEFT", "RIGHT", "UP", "DOWN"]
M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """Independent pure-Python Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011)."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def mix32(x):
    """triple32 integer hash (include/ssd_hip.h)."""
    x ^= x >> 17; x = (x * 0xED5AD4BB) & M32
    x ^= x >> 11; x = (x * 0xAC4C1B51) & M32
    x ^= x >> 15; x = (x * 0x31848BAB) & M32
    x ^= x >> 14
    return x


_BASE_CACHE = {}


def ctr_u32(seed, env, episode, c, stream, k):
    """x(stream, k) of the COUNTER-mode generator (include/ssd_hip.h) for call index c of episode `episode`;
    independent pure-Python statement."""
    key = (seed, env, episode)
    if key not in _BASE_CACHE:
        if len(_BASE_CACHE) > 4096:
            _BASE_CACHE.clear()
        _BASE_CACHE[key] = philox4x32_10((0, 0, env & M32, episode & M32), (seed & M32, (seed >> 32) & M32))
    return mix32(_BASE_CACHE[key][stream] ^ ((c << 16) & M32) ^ (k & M32))


def import_reference():
    """Appendix C of SURVEY.md: stub the two absent packages (cv2: video writer only; pyclustering: learner only)."""
    sys.dont_write_bytecode = True
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    for m in ("pyclustering", "pyclustering.cluster", "pyclustering.cluster.xmeans",
              "pyclustering.cluster.center_initializer"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import matplotlib
    matplotlib.use("Agg")
    with contextlib.redirect_stdout(io.StringIO()):
        import envs  # noqa: F401  (reference package)
    return sys.modules["envs"]


class CallRecord:
    def __init__(self):
        self.move_order = None      # list of agent indices (shuffled movers) or None
        self.uniforms = []          # f64 in consumption order
        self.waste_order = None     # list of site indices or None
        self.spawn_rot = []
        self.spawn_order = []       # random_spawn_point: per agent, the spawn point ids after its shuffle


class RefEnv:
    def __init__(self, env, map, num_agents, view_size=7, episode_limit=100, extra_args=None, mode="tape",
                 seed=0, env_id=0):
        envs = import_reference()
        ea = dict(random_spawn_point=False, random_spawn_rotation=0, disable_rotation_action=True,
                  disable_fire_action=True, obs_color="simplified")
        if extra_args:
            ea.update(extra_args)
        with contextlib.redirect_stdout(io.StringIO()):
            self.env = envs.REGISTRY[env](num_agents=num_agents, render=False, episode_limit=episode_limit,
                                          is_replay=False, view_size=view_size, map=map, extra_args=ea, seed=seed)
        self.kind, self.n, self.mode, self.seed, self.env_id = env, num_agents, mode, seed, env_id
        self.epoch = 0
        self.call_index = 0
        self.force_move_order = None
        self.H, self.W = self.env.world_map.shape
        if env == "cleanup":
            self.waste_index = {tuple(p): i for i, p in enumerate(self.env.waste_start_points)}
        else:
            self.waste_index = {}
        # spawn point ids = row-major scan order (map_env.py:140-146); the live list may already be shuffled by the constructor
        self.spawn_index = {p: i for i, p in enumerate(sorted({tuple(q) for q in self.env.spawn_points}))}
        self.n_spawn = len(self.spawn_index)             # distinct points
        self.spawn_len = len(self.env.spawn_points)      # list length: Cleanup appends every point twice (cleanup.py:79-80)
        self.n_waste = len(self.waste_index)
        self.n_apple = len(self.env.apple_points)
        self.rec = None

    # ---- RNG interception -------------------------------------------------------------------------------
    @contextlib.contextmanager
    def _intercept(self):
        rec = self.rec = CallRecord()
        o_shuffle, o_rand, o_randint, o_pyshuffle = np.random.shuffle, np.random.rand, np.random.randint, random.shuffle
        counter = self.mode == "counter"
        u32 = lambda stream, k: ctr_u32(self.seed, self.env_id, self.epoch, self.call_index, stream, k)

        def np_shuffle(lst):
            if self.force_move_order is not None:   # KAT generation: impose a chosen shuffle result
                by = {int(a.split("-")[1]): (a, sl) for a, sl in lst}
                assert sorted(by) == sorted(self.force_move_order), (sorted(by), self.force_move_order)
                lst[:] = [by[i] for i in self.force_move_order]
            elif counter:
                idx = [int(a.split("-")[1]) for a, _ in lst]
                order = sorted(range(len(lst)), key=lambda i: (u32(STREAM_MOVE, idx[i]) >> 8, idx[i]))
                lst[:] = [lst[i] for i in order]
            else:
                o_shuffle(lst)
            assert rec.move_order is None
            rec.move_order = [int(a.split("-")[1]) for a, _ in lst]

        def np_rand(*shape):
            assert shape == (1,)
            if counter:
                v = np.array([(u32(STREAM_UNIFORM, len(rec.uniforms)) >> 8) * 2.0 ** -24])
            else:
                v = o_rand(1)
            rec.uniforms.append(float(v[0]))
            return v

        def np_randint(high):
            assert high == 4
            v = (u32(STREAM_SPAWN_ROT, len(rec.spawn_rot)) >> 30) if counter else int(o_randint(high))
            rec.spawn_rot.append(v)
            return v

        def py_shuffle(lst):
            if lst is getattr(self.env, "spawn_points", None):      # spawn_point() of agent a (map_env.py:776-777)
                a = len(rec.spawn_order)
                if counter:
                    seen, eid = {}, []                   # element id = copy * n_points + point id (equal points are interchangeable)
                    for p in lst:
                        sidx = self.spawn_index[tuple(p)]
                        eid.append(seen.get(sidx, 0) * self.n_spawn + sidx)
                        seen[sidx] = seen.get(sidx, 0) + 1
                    order = sorted(range(len(lst)), key=lambda i: (u32(STREAM_SPAWN_ROT, 256 + 32 * a + eid[i]) >> 8, eid[i]))
                    lst[:] = [lst[i] for i in order]
                else:
                    o_pyshuffle(lst)
                rec.spawn_order.append([self.spawn_index[tuple(p)] for p in lst])
                return
            assert lst is getattr(self.env, "waste_points", None)
            if counter:
                idx = [self.waste_index[tuple(p)] for p in lst]
                order = sorted(range(len(lst)), key=lambda i: (u32(STREAM_WASTE, idx[i]) >> 8, idx[i]))
                lst[:] = [lst[i] for i in order]
            else:
                o_pyshuffle(lst)
            assert rec.waste_order is None
            rec.waste_order = [self.waste_index[tuple(p)] for p in lst]

        np.random.shuffle, np.random.rand, np.random.randint, random.shuffle = np_shuffle, np_rand, np_randint, py_shuffle
        try:
            yield rec
        finally:
            np.random.shuffle, np.random.rand, np.random.randint, random.shuffle = o_shuffle, o_rand, o_randint, o_pyshuffle

    # ---- env API ------------------------------------------------------------------------------------------
    def reset(self):
        self.epoch += 1                 # episode counter of the COUNTER contract
        self.call_index = 0
        with self._intercept() as rec:
            self.env.reset()
        return rec

    def step(self, actions):
        self.call_index = self.env._episode_steps + 1
        with self._intercept() as rec:
            reward, terminated, info = self.env.step(list(actions))
        # the reference aliases the first returned reward array with self.rewards and mutates it later
        # (map_env.py:885-888): hand out a copy
        return np.array(reward, copy=True), bool(terminated), info, rec

    def grid(self):
        wm = self.env.world_map
        out = np.zeros(wm.shape, np.uint8)
        for ch, code in CODES.items():
            out[wm == ch] = code
        assert set(np.unique(wm)) <= set(CODES), np.unique(wm)
        return out

    def pos(self):
        return np.array([self.env.agents["agent-%d" % i].get_pos().tolist() for i in range(self.n)], np.int16)

    def orient(self):
        return np.array([ORI[self.env.agents["agent-%d" % i].get_orientation()] for i in range(self.n)], np.uint8)

    def set_state(self, grid=None, pos=None, orient=None):
        """Agent.set_pos / set_orientation / direct world_map writes, for KATs."""
        if grid is not None:
            inv = {v: k for k, v in CODES.items()}
            for r in range(self.H):
                for c in range(self.W):
                    self.env.world_map[r, c] = inv[int(grid[r, c])]
        for i in range(self.n):
            ag = self.env.agents["agent-%d" % i]
            if pos is not None:
                ag.set_pos([int(pos[i][0]), int(pos[i][1])])
            if orient is not None:
                ag.set_orientation(ORI_NAMES[int(orient[i])])

    def obs(self):
        return np.stack(self.env.get_obs())          # f64 [n,3,V,V]

    def state(self):
        return self.env.get_state()                  # f64 [3,H,W]


def tape_arrays(rec, n_agents, max_uniforms, n_waste, n_spawn=0):
    """CallRecord -> fixed-shape arrays in the ssd_tape layout (one env)."""
    mo = np.full(n_agents, 0xFF, np.uint8)
    if rec.move_order is not None:
        mo[:len(rec.move_order)] = rec.move_order
    u = np.zeros(max(1, max_uniforms), np.float64)
    assert len(rec.uniforms) <= max(1, max_uniforms), (len(rec.uniforms), max_uniforms)
    u[:len(rec.uniforms)] = rec.uniforms
    wo = np.full(max(1, n_waste), 0xFF, np.uint8)
    if rec.waste_order is not None:
        wo[:n_waste] = rec.waste_order
    sr = np.zeros(n_agents, np.uint8)
    sr[:len(rec.spawn_rot)] = rec.spawn_rot
    so = np.full((n_agents, max(1, n_spawn)), 0xFF, np.uint8)
    for a, order in enumerate(rec.spawn_order):
        so[a, :len(order)] = order
    return dict(move_order=mo, uniforms=u, waste_order=wo, spawn_rot=sr, spawn_order=so, n_uniforms=len(rec.uniforms),
                shuffled=rec.waste_order is not None)
