"""Vectorised cadence: with batch_size_run > 1 the two schedules the reference counts in env steps / episodes of its
single env (epsilon_anneal_time, config/algs/homophily.yaml; target_update_interval, homophily_learner.py:255-257) are counted in
rollouts / learner.train calls, so the target net is NOT re-synced after every gradient step and epsilon does not collapse to its
floor after the first rollout.  batch_size_run == 1 keeps the reference's literal arithmetic."""
from types import SimpleNamespace

import numpy as np

from homophily_marl_amd.components.epsilon_schedules import DecayThenFlatSchedule
from homophily_marl_amd.runners.hip_vec_runner import HipVecRunner
from tests.learner_util import build, load_fixture


def _runner(unit, N, t_env, rollouts, T=100):
    r = HipVecRunner.__new__(HipVecRunner)
    r.args = SimpleNamespace(schedule_unit=unit)
    r.batch_size, r.episode_limit, r.t_env, r.rollouts = N, T, t_env, rollouts
    return r


def test_epsilon_clock_units():
    sched = DecayThenFlatSchedule(1.0, 0.05, 50000, decay="linear")
    # reference arithmetic, one env: t_env after k rollouts = 100 k
    assert _runner("env_steps", 1, 300, 3).sched_t == 300
    # 4096 envs, literal env steps: the floor is reached after ONE rollout (409 600 > 50 000)
    assert sched.eval(_runner("env_steps", 4096, 409600, 1).sched_t) == 0.05
    # rollouts unit: the clock advances by episode_limit per rollout, whatever the batch is -> 500 rollouts per anneal, as in the reference
    assert _runner("rollouts", 4096, 409600, 1).sched_t == 100
    assert abs(sched.eval(_runner("rollouts", 4096, 409600 * 250, 250).sched_t) - 0.525) < 1e-12
    assert abs(sched.eval(_runner("rollouts", 4096, 409600 * 500, 500).sched_t) - 0.05) < 1e-12


def test_target_net_is_not_synced_on_consecutive_train_calls():
    z, meta = load_fixture("learner_cleanup5.npz")
    args, batch, mac, learner = build(z, meta)
    syncs = []
    orig = learner._update_targets
    learner._update_targets = lambda: (syncs.append(1), orig())[1]
    # "rollouts" unit: the driver passes the number of learner.train calls so far (run.py train_iteration)
    for k in range(41):
        before = len(syncs)
        learner.train(batch, 409600 * (k + 1), k)
        if k in (20, 40):
            assert len(syncs) == before + 1, k          # every target_update_interval = 20 train steps
        else:
            assert len(syncs) == before, k
        if k == 2:
            # between syncs the target net lags the live net
            lag = max((a - b).abs().max().item() for a, b in zip(mac.agent.state_dict().values(), learner.target_mac.agent.state_dict().values()))
            assert lag > 0
    # the literal unit at 4096 envs would sync on EVERY call: episode += 4096 per iteration vs an interval of 20
    calls = sum(1 for k in range(1, 6) if (4096 * k - 4096 * (k - 1)) / args.target_update_interval >= 1.0)
    assert calls == 5


def test_setup_picks_the_unit_from_the_batch_size():
    from homophily_marl_amd import run
    src = open(run.__file__).read()
    assert 'args.schedule_unit = "env_steps" if args.batch_size_run == 1 else "rollouts"' in src
