"""Generate the committed golden fixtures under tests/golden/ FROM THE IMPORTED REFERENCE.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference).  Usage: python oracle/gen_golden.py
A fixture is data only: inputs (actions, recorded random draws, teleports) and the reference's outputs.

traj_<name>.npz  one env, a list of calls c = 0..C-1 (kind 0 = reset, 1 = step):
    meta                json: env, map, num_agents, view_size, episode_limit, extra_args
    kind[C] u8, actions[C,n] i32
    pre_pos[C,n,2] i16, pre_orient[C,n] u8      agent state fed to the call (after an optional teleport)
    move_order[C,n] u8, uniforms[C,U] f64, n_uniforms[C] i32, waste_order[C,Wn] u8, shuffled[C] u8, spawn_rot[C,n] u8
    grid[C,H,W] u8, pos[C,n,2] i16, orient[C,n] u8                     state after the call
    reward[C,n] f64, clean_num[C,n] f64, apple_den[C] f64, terminated[C] u8, collective_return[C] f64, equality[C] f64
    obs_calls[K] i32, obs_u8[K,n,3,V,V] u8 (get_obs()*256), state_u8[K,3,H,W] u8, agent_pos[K,n,2] f64, agent_orient[K,n,2] f64
kat_moves.npz    move-conflict known-answer tests: for each scenario every shuffle result of the movers is imposed on
                 the reference (np.random.shuffle forced) and the resulting positions recorded.
"""
import itertools
import json
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ref_harness as RH  # noqa: E402
from oracle.check_vs_reference import cluster_positions  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
ALL = dict(disable_rotation_action=False, disable_fire_action=False)

TRAJ = [
    # name, cfg, episodes, episode_limit, extra_args, seed, teleport prob
    ("cleanup3_default", dict(env="cleanup", map="default3", num_agents=3, view_size=7), 2, 50, None, 11, 0.1),
    ("cleanup3_allact", dict(env="cleanup", map="default3", num_agents=3, view_size=7), 2, 50, ALL, 12, 0.1),
    ("cleanup5_default", dict(env="cleanup", map="default5", num_agents=5, view_size=7), 1, 100, None, 21, 0.0),
    ("cleanup5_cluster", dict(env="cleanup", map="default5", num_agents=5, view_size=7), 2, 50, None, 22, 0.2),
    ("cleanup5_allact", dict(env="cleanup", map="default5", num_agents=5, view_size=7), 2, 50, ALL, 23, 0.15),
    ("cleanup5_randrot_full", dict(env="cleanup", map="default5", num_agents=5, view_size=7), 2, 40,
     dict(ALL, random_spawn_rotation=None, obs_color="full"), 24, 0.15),
    ("cleanup10_default", dict(env="cleanup", map="default10", num_agents=10, view_size=7), 1, 100, None, 31, 0.1),
    ("cleanup10_allact", dict(env="cleanup", map="default10", num_agents=10, view_size=7), 2, 40, ALL, 32, 0.2),
    ("harvest5_default", dict(env="harvest", map="default10", num_agents=5, view_size=15), 1, 100, None, 41, 0.1),
    ("harvest5_allact", dict(env="harvest", map="default10", num_agents=5, view_size=15), 2, 40,
     dict(ALL, random_spawn_rotation=None), 42, 0.15),
    ("harvest10_v7_allact", dict(env="harvest", map="default10", num_agents=10, view_size=7), 2, 40,
     dict(ALL, obs_color="full"), 43, 0.2),
    # random_spawn_point: every agent's spawn_point() shuffles the spawn list first (map_env.py:776-777); 3 resets each
    ("cleanup5_randspawn", dict(env="cleanup", map="default5", num_agents=5, view_size=7), 3, 20,
     dict(random_spawn_point=True, random_spawn_rotation=None), 51, 0.1),
    ("harvest5_randspawn", dict(env="harvest", map="default10", num_agents=5, view_size=7), 3, 20,
     dict(ALL, random_spawn_point=True), 52, 0.1),
]


def gen_traj(name, cfg, episodes, limit, ea, seed, p_tele):
    np.random.seed(seed); random.seed(seed)
    rng = np.random.default_rng(seed)
    ref = RH.RefEnv(cfg["env"], cfg["map"], cfg["num_agents"], cfg["view_size"], limit, ea)
    n = cfg["num_agents"]
    n_actions = ref.env.n_actions
    maxu, nw = ref.n_apple + ref.n_waste, ref.n_waste
    full = ea is not None and not ea.get("disable_rotation_action", True)
    avail = list(range(n_actions)) if full else [a for a in range(n_actions) if a not in (5, 6, 7)]
    rows = []
    obs_rows = []

    def add(kind, acts, pre_pos, pre_orient, rec, reward, info, term):
        ta = RH.tape_arrays(rec, n, maxu, nw, ref.spawn_len)
        rsp = bool(ea and ea.get("random_spawn_point"))
        rows.append(dict(
            **({"spawn_order": ta["spawn_order"]} if rsp else {}),
            kind=kind, actions=acts, pre_pos=pre_pos, pre_orient=pre_orient,
            move_order=ta["move_order"], uniforms=ta["uniforms"], n_uniforms=ta["n_uniforms"],
            waste_order=ta["waste_order"], shuffled=int(ta["shuffled"]), spawn_rot=ta["spawn_rot"],
            grid=ref.grid(), pos=ref.pos(), orient=ref.orient(),
            # copies: the reference keeps aliasing the first step's reward array (map_env.py:885-888)
            reward=np.zeros(n) if reward is None else np.array(reward, copy=True),
            clean_num=np.zeros(n) if info is None else np.array(info["clean_num"], copy=True),
            apple_den=0.0 if info is None else float(info["apple_den"][0]),
            terminated=int(term),
            collective_return=float(info["collective_return"]) if term else np.nan,
            equality=float(info["equality_metric"]) if term else np.nan))

    def add_obs():
        o = ref.obs() * 256
        s = ref.state() * 256
        assert (o == np.round(o)).all() and o.min() >= 0 and o.max() <= 255
        obs_rows.append(dict(call=len(rows) - 1, obs=o.astype(np.uint8), state=s.astype(np.uint8),
                             agent_pos=ref.env.get_agent_pos(), agent_orient=ref.env.get_agent_orientation()))

    for ep in range(episodes):
        pre_pos = ref.pos() if ep else np.zeros((n, 2), np.int16)
        pre_ori = ref.orient() if ep else np.zeros(n, np.uint8)
        rec = ref.reset()
        add(0, np.zeros(n, np.int32), pre_pos, pre_ori, rec, None, None, False)
        add_obs()
        term = False
        t = 0
        while not term:
            if rng.random() < p_tele:
                p = cluster_positions(rng, ref.grid(), n, int(rng.integers(1, 3)))
                ori = rng.integers(0, 4, n).astype(np.uint8) if full else ref.orient()
                ref.set_state(pos=p, orient=ori)
            pre_pos, pre_ori = ref.pos(), ref.orient()
            acts = np.array([rng.choice(avail) for _ in range(n)], np.int32)
            reward, term, info, rec = ref.step(acts)
            add(1, acts, pre_pos, pre_ori, rec, reward, info, term)
            t += 1
            if t % 17 == 3 or term:
                add_obs()
    out = {k: np.stack([np.asarray(r[k]) for r in rows]) for k in rows[0]}
    out["kind"] = out["kind"].astype(np.uint8); out["terminated"] = out["terminated"].astype(np.uint8)
    out["shuffled"] = out["shuffled"].astype(np.uint8); out["n_uniforms"] = out["n_uniforms"].astype(np.int32)
    out["uniforms"] = out["uniforms"][:, :max(1, int(out["n_uniforms"].max()))]
    out["obs_calls"] = np.array([r["call"] for r in obs_rows], np.int32)
    out["obs_u8"] = np.stack([r["obs"] for r in obs_rows])
    out["state_u8"] = np.stack([r["state"] for r in obs_rows])
    out["agent_pos"] = np.stack([r["agent_pos"] for r in obs_rows])
    out["agent_orient"] = np.stack([r["agent_orient"] for r in obs_rows])
    meta = dict(cfg, episode_limit=limit, extra_args=ea, n_actions=int(n_actions), n_apple=ref.n_apple, n_waste=ref.n_waste,
                generator="oracle/gen_golden.py", seed=seed)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    path = os.path.join(OUT, "traj_%s.npz" % name)
    np.savez_compressed(path, **out)
    ev = dict(calls=len(rows), eaten=int((out["reward"] > 0).sum()), cleaned=int(out["clean_num"].sum()),
              shuffles=int(out["shuffled"].sum()), kb=os.path.getsize(path) // 1024)
    print("wrote", path, ev, flush=True)


def gen_kats():
    """Move-conflict KATs on Cleanup default5 (the A.7 cases of SURVEY.md plus random tight clusters)."""
    np.random.seed(7); random.seed(7)
    rng = np.random.default_rng(7)
    ref = RH.RefEnv("cleanup", "default5", 5, 7, 1000, ALL)
    ref.reset()
    base_grid = ref.grid()
    scen = [
        ([(10, 10), (10, 9), (10, 8), (9, 9), (20, 10)], [0, 0, 0, 0, 0], [4, 0, 0, 3, 4]),   # contested + blocked occupant
        ([(10, 8), (10, 9), (9, 9), (9, 8), (20, 10)], [0, 0, 0, 0, 0], [0, 2, 1, 3, 4]),     # 4-cycle
        ([(10, 8), (10, 9), (15, 9), (16, 9), (20, 10)], [0, 0, 0, 0, 0], [0, 1, 4, 4, 4]),   # swap
        ([(10, 8), (10, 9), (10, 10), (16, 9), (20, 10)], [0, 0, 0, 0, 0], [0, 0, 0, 4, 4]),  # train
        ([(10, 8), (10, 10), (9, 9), (11, 9), (20, 10)], [0, 0, 0, 0, 0], [0, 1, 3, 2, 4]),   # 4 agents, one cell
        ([(10, 9), (10, 9), (10, 10), (9, 9), (20, 10)], [0, 0, 0, 0, 0], [0, 1, 1, 3, 4]),   # start with a shared cell
    ]
    for _ in range(60):
        p = cluster_positions(rng, base_grid, 5, int(rng.integers(1, 3)))
        scen.append(([tuple(x) for x in p.tolist()], rng.integers(0, 4, 5).tolist(), rng.integers(0, 7, 5).tolist()))
    S_pos, S_ori, S_act, S_perm, S_out, S_oori = [], [], [], [], [], []
    for pos, ori, acts in scen:
        movers = [i for i, a in enumerate(acts) if a <= 4]
        perms = list(itertools.permutations(movers)) if movers else [()]
        for pm in perms:
            ref.set_state(grid=base_grid, pos=np.array(pos), orient=np.array(ori))
            ref.force_move_order = list(pm)
            ref.step(acts)
            ref.force_move_order = None
            mo = np.full(5, 0xFF, np.uint8); mo[:len(pm)] = pm
            S_pos.append(pos); S_ori.append(ori); S_act.append(acts); S_perm.append(mo)
            S_out.append(ref.pos()); S_oori.append(ref.orient())
    path = os.path.join(OUT, "kat_moves.npz")
    np.savez_compressed(path, grid=base_grid, pos=np.array(S_pos, np.int16), orient=np.array(S_ori, np.uint8),
                        actions=np.array(S_act, np.int32), move_order=np.array(S_perm, np.uint8),
                        out_pos=np.array(S_out, np.int16), out_orient=np.array(S_oori, np.uint8))
    sh = sum(len({tuple(q) for q in o.tolist()}) < 5 for o in S_out)
    print("wrote", path, "cases:", len(S_pos), "ending with a shared cell:", sh, "kb", os.path.getsize(path) // 1024)


if __name__ == "__main__":
    import sys
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])                  # optional: names of the trajectories to (re)generate
    for row in TRAJ:
        if not only or row[0] in only:
            gen_traj(*row)
    if not only or "kats" in only:
        gen_kats()
