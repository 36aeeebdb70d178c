"""Property test: C restatement (oracle/ssd_oracle.c) == imported reference, step by step.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference).  Usage:
    python oracle/check_vs_reference.py [--steps 12000] [--seed 0]
Every step compares grid, positions, orientations, reward, clean_num, apple_den, terminated, the number of uniforms
consumed, and (every --obs-every steps) the full get_obs() / get_state() tensors.  Agents are periodically
teleported into a tight cluster (Agent.set_pos on the reference, ssd_cpu_import_state on the oracle) so the
move-conflict code of map_env.py:553-661 is exercised far more often than a random walk would.
Both RNG contracts are covered: mode=tape (unmodified reference RNG, recorded) and mode=counter (reference RNG
functions replaced by the Philox counter generator).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from homophily_marl_amd import abi  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402
from oracle.oracle_py import OracleEnv, make_tape  # noqa: E402

CONFIGS = [
    dict(env="cleanup", map="default3", num_agents=3, view_size=7),
    dict(env="cleanup", map="default5", num_agents=5, view_size=7),
    dict(env="cleanup", map="default10", num_agents=10, view_size=7),
    dict(env="harvest", map="default10", num_agents=5, view_size=15),
    dict(env="harvest", map="default10", num_agents=10, view_size=7),
]


def cluster_positions(rng, grid, n, spread):
    """n positions (duplicates allowed with small probability) on non-wall cells near a random centre."""
    H, W = grid.shape
    free = np.argwhere(grid != 1)
    centre = free[rng.integers(len(free))]
    out = []
    while len(out) < n:
        p = centre + rng.integers(-spread, spread + 1, 2)
        if 0 <= p[0] < H and 0 <= p[1] < W and grid[p[0], p[1]] != 1:
            if any((p == q).all() for q in out) and rng.random() > 0.1:
                continue
            out.append(p)
    return np.array(out, np.int16)


def run_config(cfg, steps, seed, mode, extra_args, obs_every, episode_limit=50):
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    import random
    random.seed(seed)
    ref = RH.RefEnv(cfg["env"], cfg["map"], cfg["num_agents"], cfg["view_size"], episode_limit, extra_args,
                    mode=mode, seed=seed * 7919 + 13, env_id=3)
    rmode = abi.RNG_TAPE if mode == "tape" else abi.RNG_COUNTER
    orc = OracleEnv(cfg["env"], map=cfg["map"], num_agents=cfg["num_agents"], n_env=1, view_size=cfg["view_size"],
                    episode_limit=episode_limit, extra_args=extra_args, rng_mode=rmode, seed=seed * 7919 + 13,
                    env_id_base=3)
    n = cfg["num_agents"]
    maxu, nw = orc.info.max_uniforms, orc.info.n_waste_sites
    assert orc.info.n_apple_sites == ref.n_apple and nw == ref.n_waste
    full_actions = extra_args is not None and not extra_args.get("disable_rotation_action", True)
    avail = list(range(orc.n_actions)) if full_actions else [a for a in range(orc.n_actions) if a not in (5, 6, 7)]

    def tape_of(rec):
        ta = RH.tape_arrays(rec, n, maxu, nw, ref.spawn_len)
        t, _ = make_tape(1, n, maxu, nw, ta["move_order"][None], ta["uniforms"][None], ta["waste_order"][None],
                         ta["spawn_rot"][None], ta["spawn_order"][None])
        return t, ta

    def compare(tag, t):
        st = orc.export_state()
        g = ref.grid()
        assert (st["grid"][0].reshape(g.shape) == g).all(), (tag, t, "grid")
        assert (st["pos"][0] == ref.pos()).all(), (tag, t, "pos", st["pos"][0].tolist(), ref.pos().tolist())
        assert (st["orient"][0] == ref.orient()).all(), (tag, t, "orient")

    def compare_obs(t):
        o = orc.observe(abi.OBS_F32, want_state=True)
        ro = ref.obs()
        assert (o["obs"][0] == ro.astype(np.float32)).all(), (t, "obs")
        assert (o["obs"][0].astype(np.float64) == ro).all(), (t, "obs f64")
        assert (o["state"][0].astype(np.float64) == ref.state()).all(), (t, "state")
        assert (o["pos"][0].astype(np.float64) == ref.env.get_agent_pos()).all()
        assert (o["orient"][0].astype(np.float64) == ref.env.get_agent_orientation()).all()
        if (extra_args or {}).get("obs_color", "simplified") == "simplified":
            oc = orc.observe(abi.OBS_CODE)["obs"][0]
            u8 = orc.observe(abi.OBS_U8)["obs"][0]
            code = np.where(u8[:, 1] > 0, 1, np.where(u8[:, 0] > 0, 2, np.where(u8[:, 2] > 0, 3, 0)))
            assert (oc == code).all()

    t_done = 0
    while t_done < steps:
        rec = ref.reset()
        t, ta = tape_of(rec)
        o = orc.reset(t)
        assert o["n_draws"][0] == ta["n_uniforms"], ("reset draws", o["n_draws"][0], ta["n_uniforms"])
        compare("reset", t_done)
        compare_obs(t_done)
        term = False
        ep_t = 0
        while not term:
            if rng.random() < 0.15:  # teleport into a cluster
                g = ref.grid()
                p = cluster_positions(rng, g, n, spread=int(rng.integers(1, 3)))
                ori = rng.integers(0, 4, n).astype(np.uint8) if full_actions else ref.orient()
                ref.set_state(pos=p, orient=ori)
                orc.import_state(pos=p[None], orient=ori[None])
            p_stay = rng.random()
            acts = np.array([rng.choice(avail) if rng.random() > 0.2 * p_stay else 4 for _ in range(n)], np.int32)
            reward, term, info, rec = ref.step(acts)
            t, ta = tape_of(rec)
            o = orc.step(acts[None], t)
            assert o["n_draws"][0] == ta["n_uniforms"], ("step draws", t_done, o["n_draws"][0], ta["n_uniforms"])
            compare("step", t_done)
            assert (o["reward"][0].astype(np.float64) == reward).all(), (t_done, "reward", o["reward"][0], reward)
            assert (o["clean_num"][0].astype(np.float64) == info["clean_num"]).all(), (t_done, "clean_num")
            assert (o["apple_den"][0] == info["apple_den"].astype(np.float32)).all(), (t_done, "apple_den")
            assert bool(o["terminated"][0]) == term
            if term:
                assert o["collective_return"][0] == np.float32(info["collective_return"])
                assert o["equality"][0] == np.float32(info["equality_metric"]), (o["equality"][0], info["equality_metric"])
            ep_t += 1
            t_done += 1
            if t_done % obs_every == 0:
                compare_obs(t_done)
    return t_done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000, help="steps per (config, mode, option set)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--obs-every", type=int, default=7)
    a = ap.parse_args()
    option_sets = [
        None,                                                            # defaults: masked turn/fire, spawn rot 0 (LEFT)
        dict(disable_rotation_action=False, disable_fire_action=False),  # all actions
        dict(disable_rotation_action=False, disable_fire_action=False, random_spawn_rotation=None, obs_color="full"),
        dict(random_spawn_rotation=3),
        dict(random_spawn_point=True, random_spawn_rotation=None, disable_rotation_action=False, disable_fire_action=False),
    ]
    total = 0
    t0 = time.time()
    for cfg in CONFIGS:
        for mode in ("tape", "counter"):
            for oi, ea in enumerate(option_sets):
                nsteps = run_config(cfg, a.steps, a.seed + oi, mode, ea, a.obs_every)
                total += nsteps
                print("ok  %-8s %-10s n=%-2d mode=%-7s opts=%d  %d steps  (%.0fs)" %
                      (cfg["env"], cfg["map"], cfg["num_agents"], mode, oi, nsteps, time.time() - t0), flush=True)
    print("ALL OK: %d steps compared, 0 mismatches" % total)


if __name__ == "__main__":
    main()
