"""Replay of the committed golden fixtures (tests/golden/, generated from the reference by oracle/gen_golden.py)
through any backend with the numpy interface of oracle.oracle_py.OracleEnv (the C oracle, or the HIP adapter)."""
import glob
import json
import os

import numpy as np

from homophily_marl_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def traj_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


def load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def env_kwargs(meta, n_env=1, rng_mode=abi.RNG_TAPE, **kw):
    return dict(map=meta["map"], num_agents=meta["num_agents"], n_env=n_env, view_size=meta["view_size"],
                episode_limit=meta["episode_limit"], extra_args=meta["extra_args"], rng_mode=rng_mode, **kw)


def replay(make_env, make_tape, path, n_env=1):
    """make_env(env_name, **kwargs) -> backend; make_tape(n_env, n, U, Wn, mo, un, wo, sr[, spawn_order]) -> SsdTape-like accepted by
    the backend.  With n_env > 1 the same trajectory is replicated in every env (checks batch indexing)."""
    z, meta = load(path)
    n = meta["num_agents"]
    env = make_env(meta["env"], **env_kwargs(meta, n_env))
    U = z["uniforms"].shape[1]
    Wn = z["waste_order"].shape[1]
    rep = lambda a: np.repeat(np.asarray(a)[None], n_env, 0)
    obs_at = {int(c): k for k, c in enumerate(z["obs_calls"])}
    for c in range(len(z["kind"])):
        tape = make_tape(n_env, n, U, Wn, rep(z["move_order"][c]), rep(z["uniforms"][c]), rep(z["waste_order"][c]),
                         rep(z["spawn_rot"][c]), *((rep(z["spawn_order"][c]),) if "spawn_order" in z else ()))
        if z["kind"][c] == 0:
            o = env.reset(tape)
        else:
            env.import_state(pos=rep(z["pre_pos"][c]), orient=rep(z["pre_orient"][c]))
            o = env.step(rep(z["actions"][c]), tape)
        st = env.export_state()
        for e in range(n_env):
            tag = (os.path.basename(path), c, e)
            assert o["n_draws"][e] == z["n_uniforms"][c], tag + ("n_draws", int(o["n_draws"][e]), int(z["n_uniforms"][c]))
            assert (st["grid"][e] == z["grid"][c].reshape(-1)).all(), tag + ("grid",)
            assert (st["pos"][e] == z["pos"][c]).all(), tag + ("pos",)
            assert (st["orient"][e] == z["orient"][c]).all(), tag + ("orient",)
            if z["kind"][c] == 1:
                assert (o["reward"][e] == z["reward"][c].astype(np.float32)).all(), tag + ("reward",)
                assert (o["clean_num"][e] == z["clean_num"][c].astype(np.float32)).all(), tag + ("clean_num",)
                assert (o["apple_den"][e] == np.float32(z["apple_den"][c])).all(), tag + ("apple_den",)
                assert o["terminated"][e] == z["terminated"][c], tag + ("terminated",)
                if z["terminated"][c]:
                    assert o["collective_return"][e] == np.float32(z["collective_return"][c]), tag
                    assert o["equality"][e] == np.float32(z["equality"][c]), tag
        if c in obs_at:
            k = obs_at[c]
            ob = env.observe(abi.OBS_U8, want_state=True)
            of = env.observe(abi.OBS_F32)
            for e in range(n_env):
                tag = (os.path.basename(path), c, e)
                assert (ob["obs"][e] == z["obs_u8"][k]).all(), tag + ("obs u8",)
                assert (of["obs"][e] == (z["obs_u8"][k].astype(np.float64) / 256).astype(np.float32)).all(), tag + ("obs f32",)
                assert (ob["state"][e] == (z["state_u8"][k].astype(np.float64) / 256).astype(np.float32)).all(), tag + ("state",)
                assert (ob["pos"][e] == z["agent_pos"][k].astype(np.float32)).all(), tag + ("agent_pos",)
                assert (ob["orient"][e] == z["agent_orient"][k].astype(np.float32)).all(), tag + ("agent_orient",)
            if (meta["extra_args"] or {}).get("obs_color", "simplified") == "simplified":
                u8 = z["obs_u8"][k]
                code = np.where(u8[:, 1] > 0, 1, np.where(u8[:, 0] > 0, 2, np.where(u8[:, 2] > 0, 3, 0))).astype(np.uint8)
                oc = env.observe(abi.OBS_CODE)
                bf = env.observe(abi.OBS_BF16)
                for e in range(n_env):
                    assert (oc["obs"][e] == code).all(), (os.path.basename(path), c, e, "obs code")
                    f = (bf["obs"][e].astype(np.uint32) << 16).view(np.float32)
                    assert (f == of["obs"][e]).all(), (os.path.basename(path), c, e, "obs bf16")
    env.close()
    return len(z["kind"])


def replay_kats(make_env, make_tape, chunk=512):
    """kat_moves.npz: every shuffle result of every scenario, batched `chunk` envs at a time."""
    z = np.load(os.path.join(GOLDEN, "kat_moves.npz"))
    N = len(z["pos"])
    n = z["pos"].shape[1]
    done = 0
    for s in range(0, N, chunk):
        m = min(chunk, N - s)
        env = make_env("cleanup", map="default5", num_agents=n, n_env=m, view_size=7, episode_limit=1000,
                       extra_args=dict(disable_rotation_action=False, disable_fire_action=False), rng_mode=abi.RNG_TAPE)
        U, Wn = env.info.max_uniforms, env.info.n_waste_sites
        t0 = make_tape(m, n, U, Wn, None, np.full((m, U), 0.999), np.tile(np.arange(Wn, dtype=np.uint8), (m, 1)), None)
        env.reset(t0)
        env.import_state(grid=np.tile(z["grid"].reshape(1, -1), (m, 1)), pos=z["pos"][s:s + m], orient=z["orient"][s:s + m])
        t1 = make_tape(m, n, U, Wn, z["move_order"][s:s + m], np.full((m, U), 0.999),
                       np.tile(np.arange(Wn, dtype=np.uint8), (m, 1)), None)
        env.step(z["actions"][s:s + m], t1)
        st = env.export_state()
        bad = np.nonzero((st["pos"] != z["out_pos"][s:s + m]).any(axis=(1, 2)) | (st["orient"] != z["out_orient"][s:s + m]).any(axis=1))[0]
        assert len(bad) == 0, ("KAT mismatch", (s + bad[:5]).tolist(), st["pos"][bad[0]].tolist(), z["out_pos"][s + bad[0]].tolist(),
                               z["pos"][s + bad[0]].tolist(), z["actions"][s + bad[0]].tolist(), z["move_order"][s + bad[0]].tolist())
        env.close()
        done += m
    return done
