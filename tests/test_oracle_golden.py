"""CPU suite: the C oracle (oracle/ssd_oracle.c) against the golden vectors generated from the reference."""
import numpy as np
import pytest

from homophily_marl_amd import abi
from oracle.oracle_py import OracleEnv, make_tape, philox
from tests import golden_util as GU


def _tape(*a):
    return make_tape(*a)[0]


@pytest.mark.parametrize("path", GU.traj_files(), ids=lambda p: p.split("traj_")[-1][:-4])
def test_trajectory(oracle_lib, path):
    assert GU.replay(OracleEnv, _tape, path, n_env=1) > 50


def test_trajectory_batched(oracle_lib):
    path = [p for p in GU.traj_files() if "cleanup5_cluster" in p][0]
    GU.replay(OracleEnv, _tape, path, n_env=3)


def test_move_kats(oracle_lib):
    assert GU.replay_kats(OracleEnv, _tape) > 2000


def test_philox_known_answers(oracle_lib):
    # Random123 kat_vectors for philox4x32-10: x0 of (ctr, key)
    assert philox(0, 0, 0, 0, 0) == 0x6627E8D5
    assert philox(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF) == 0x408F276D
    pi = (0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822 | (0x299F31D0 << 32))
    assert [philox(*pi, word=w) for w in range(4)] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_counter_generator_matches_its_python_statement(oracle_lib):
    # the two-level generator of include/ssd_hip.h: C oracle == independent pure-Python statement
    from oracle.oracle_py import counter_u32
    from oracle.ref_harness import ctr_u32
    for seed, env, episode in ((0, 0, 0), (0xDEADBEEFCAFE, 4097, 12345), (2 ** 64 - 1, 2 ** 32 - 1, 7)):
        for c in (0, 1, 100, 65535):
            for stream in range(4):
                for k in (0, 1, 2, 63, 64, 1000, 65535):
                    assert counter_u32(seed, env, episode, c, stream, k) == ctr_u32(seed, env, episode, c, stream, k)


def test_reset_spawn_points(oracle_lib):
    # SURVEY.md A.7 (checked against the reference): default flags -> agent-0 takes the LAST spawn point, all face LEFT
    exp = {("cleanup", "default5", 5): [(19, 6), (16, 11), (12, 11), (6, 6), (2, 11)],
           ("cleanup", "default3", 3): [(8, 3), (5, 4), (1, 6)],
           ("harvest", "default10", 5): [(7, 33), (7, 29), (7, 22), (7, 11), (7, 4)]}
    for (env, mp, n), pos in exp.items():
        e = OracleEnv(env, map=mp, num_agents=n, n_env=1, rng_mode=abi.RNG_COUNTER, seed=3)
        e.reset()
        st = e.export_state()
        assert st["pos"][0].tolist() == [list(p) for p in pos]
        assert (st["orient"][0] == 0).all()


def test_counter_mode_is_deterministic_and_sharded(oracle_lib):
    # envs are keyed by GLOBAL id: a shard [4,8) of an 8-env job reproduces envs 4..7 of the unsharded job
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 9, (20, 8, 5))
    full = OracleEnv("cleanup", map="default5", num_agents=5, n_env=8, rng_mode=abi.RNG_COUNTER, seed=99)
    shard = OracleEnv("cleanup", map="default5", num_agents=5, n_env=4, rng_mode=abi.RNG_COUNTER, seed=99, env_id_base=4)
    full.reset(); shard.reset()
    for t in range(20):
        a = full.step(acts[t]); b = shard.step(acts[t, 4:])
        assert (a["reward"][4:] == b["reward"]).all()
    assert (full.export_state()["grid"][4:] == shard.export_state()["grid"]).all()
    assert (full.export_state()["pos"][4:] == shard.export_state()["pos"]).all()
    assert not (full.export_state()["pos"][0] == full.export_state()["pos"][1]).all()


def test_invalid_arguments(oracle_lib):
    with pytest.raises(abi.SsdError):
        OracleEnv("cleanup", map="default5", num_agents=11, n_env=1)
    with pytest.raises(abi.SsdError):
        OracleEnv("cleanup", map="default3", num_agents=5, n_env=1)   # 3 spawn points only (map_env.py:783)
    from oracle.oracle_py import make_tape
    t = OracleEnv("cleanup", map="default5", num_agents=5, n_env=1, extra_args=dict(random_spawn_point=True), rng_mode=abi.RNG_TAPE)
    with pytest.raises(abi.SsdError):
        t.reset(make_tape(1, 5, t.info.max_uniforms, t.info.n_waste_sites)[0])   # random_spawn_point needs tape.spawn_order
    with pytest.raises(AttributeError):
        OracleEnv("harvest", map="default", num_agents=5, n_env=1)    # harvest.py:20-22,118
    e = OracleEnv("harvest", map="default10", num_agents=5, n_env=1, rng_mode=abi.RNG_COUNTER)
    e.reset()
    with pytest.raises(abi.SsdError):
        e.step(np.full((1, 5), 8))                                       # action 8 does not exist in Harvest
