"""CPU suite: controller inputs, Q-network, incentive transfer and the homophily learner of this package against the
fixtures generated from the reference (oracle/gen_learner_golden.py).  Tolerance: the Q-loss must match within 1e-5
in fp32 (BASELINE.json north_star); measured differences are ~1e-8 and come from GEMM summation order only."""
import numpy as np
import pytest
import torch as th

import os

from tests.learner_util import GOLDEN, build, load_fixture, param_checksums

FIXTURES = ["learner_cleanup5.npz", "learner_harvest5.npz", "learner_cleanup5_w4.npz"]      # _w4: 4 x the initial weights ("trained magnitude")
LOSS_TOL = 1e-5


@pytest.mark.parametrize("name", FIXTURES)
def test_build_inputs_matches_reference(name):
    z, meta = load_fixture(name)
    args, batch, mac, learner = build(z, meta)
    with th.no_grad():
        for t, key in ((0, "inputs_t0"), (3, "inputs_t3")):
            got = mac._build_inputs(batch, t).numpy()
            ref = z[key]
            # the non-visual tail is integer / exact arithmetic: bit-exact; the conv features differ by GEMM rounding only
            nf = args.obs_dim_net
            assert (got[:, nf:] == ref[:, nf:]).all(), key
            assert np.abs(got[:, :nf] - ref[:, :nf]).max() < 1e-6


@pytest.mark.parametrize("name", FIXTURES)
def test_q_values_match_reference(name):
    z, meta = load_fixture(name)
    args, batch, mac, learner = build(z, meta)
    with th.no_grad():
        q_env, q_inc = learner.unroll(mac, batch)
    assert np.abs(q_env.numpy() - z["q_env"]).max() < 2e-6
    assert np.abs(q_inc.numpy() - z["q_inc"]).max() < 2e-6


@pytest.mark.parametrize("name", FIXTURES)
def test_two_learner_steps_match_reference(name):
    """Losses of two consecutive steps and every parameter after each step (pins the two-Adam / double-clip order and
    the shared encoder being stepped by both optimisers, homophily_learner.py:43-44,220-226)."""
    z, meta = load_fixture(name)
    args, batch, mac, learner = build(z, meta)
    for step in range(2):
        logs = learner.cal_loss_and_step(batch)
        for k in ("loss_value_env", "loss_value_inc", "loss_sim", "value_give_mean", "value_receive_mean", "q_env_taken_mean",
                  "q_inc_taken_mean", "incentives_to_cleanup_per", "incentives_to_harvest_per"):
            ref = float(z["step%d_%s" % (step, k)])
            assert abs(float(logs[k]) - ref) < LOSS_TOL, (step, k, float(logs[k]), ref)
        sums, sqs, heads = param_checksums(mac)
        assert [str(x) for x in z["param_names"]] == list(mac.agent.state_dict().keys())
        assert np.abs(heads - z["step%d_param_head" % step]).max() < 2e-5, step
        assert np.abs(sqs - z["step%d_param_sq" % step]).max() / np.abs(z["step%d_param_sq" % step]).max() < 1e-5, step


def test_state_dict_is_interchangeable_with_the_reference():
    z, meta = load_fixture(FIXTURES[0])
    args, batch, mac, learner = build(z, meta)
    ours = {k: tuple(v.shape) for k, v in mac.agent.state_dict().items()}
    ref = {k[2:]: z[k].shape for k in z.files if k.startswith("w_")}
    assert ours == ref
    assert sum(int(np.prod(s)) for s in ours.values()) == 322638          # SURVEY.md appendix B, Cleanup-5
    assert len(learner.params_env) == 22 and len(learner.params_inc) == 22
    shared = {id(p) for p in learner.params_env} & {id(p) for p in learner.params_inc}
    assert len(shared) == 4                                                # the conv encoder is in both groups


def test_incentive_transfer_closed_form():
    from homophily_marl_amd import ops
    g = th.Generator().manual_seed(0)
    a = th.randint(0, 3, (3, 6, 4, 4), generator=g)
    r = th.randn(3, 5, 4, generator=g)
    give, rp, rn, rz, re, ri = ops.incentive_transfer(a, r, 1.0, 0.1, 1.0, 6.0)
    m = a * (1 - th.eye(4, dtype=a.dtype))
    assert (give == (m[:, :-1] != 0).sum(3)).all()
    assert (rp == (m == 1).sum(2)).all() and (rn == (m == 2).sum(2)).all() and (rz == 3 - rp - rn).all()
    assert th.equal(re, (r + (rp - rn)[:, :-1] * 1.0 * 1.0) / 6.0)
    assert th.equal(ri, (r - give * 0.1 * 1.0) / 6.0)


def test_build_inputs_non_shipped_flag_sets():
    """HomophilyMAC._build_inputs with flag combinations other than the shipped one (obs_others_last_action, obs_distance, blocks
    switched off) against the reference controller's output on the cleanup fixture batch (oracle/gen_inputs_golden.py)."""
    import json
    from types import SimpleNamespace
    from homophily_marl_amd.controllers import REGISTRY as mac_REGISTRY
    z, meta = load_fixture("learner_cleanup5.npz")
    zi = np.load(os.path.join(GOLDEN, "inputs_flags.npz"))
    flag_sets = json.loads(bytes(zi["flag_sets"]).decode())
    args, batch, _, _ = build(z, meta)
    for i, flags in enumerate(flag_sets):
        a = SimpleNamespace(**dict(vars(args), **flags))
        mac = mac_REGISTRY[a.mac](batch.scheme, {"agents": a.n_agents}, a)
        assert not mac.shipped_flags
        for t in zi["ts"]:
            ref = zi["tail_%d_t%d" % (i, t)]
            with th.no_grad():
                got = mac._build_inputs(batch, int(t))[:, a.obs_dim_net:].numpy()
            assert got.shape == ref.shape == (batch.batch_size * a.n_agents, mac.input_shape - a.obs_dim_net), (i, t, got.shape, ref.shape)
            assert np.abs(got - ref).max() < 1e-6, (i, int(t), flags)
        # the time-batched form used by the learner (previous-step features shifted in time) agrees with the stepwise one
        with th.no_grad():
            q_env, q_inc = mac.unroll(batch)
            mac.init_hidden(batch.batch_size)
            for t in range(3):
                qe, qi, _ = mac.forward(batch, t)
                assert (qe - q_env[:, t]).abs().max() < 1e-5 and (qi - q_inc[:, t]).abs().max() < 1e-5


def test_class_code_storage_expands_to_the_reference_observation():
    """obs_storage: "code": HomophilyMAC.expand_codes on the class codes reproduces, bit for bit, the observations the reference
    stored in the learner fixture (simplified palette), so the learner sees identical inputs from the compact storage."""
    from homophily_marl_amd.controllers.homophily_controller import HomophilyMAC
    z, meta = load_fixture("learner_cleanup5.npz")
    o = z["batch_obs"]                                                   # u8 [B, T, n, 3, V, V], values k of k / 256
    assert set(np.unique(o)) <= {0, 255} and (o.astype(np.int32).sum(axis=3) <= 255).all()    # at most one channel per cell
    codes = np.where(o[:, :, :, 0] == 255, 2, np.where(o[:, :, :, 1] == 255, 1, np.where(o[:, :, :, 2] == 255, 3, 0))).astype(np.uint8)
    got = HomophilyMAC.expand_codes(th.as_tensor(codes))
    assert got.shape == o.shape and th.equal(got, th.as_tensor(o).float() / 256)
    args, batch, mac, _ = build(z, meta)
    with th.no_grad():
        qe, qi = mac.unroll(batch)
        batch.data.transition_data["obs"] = th.as_tensor(codes)          # the compact form of the same batch
        qe2, qi2 = mac.unroll(batch)
    assert th.equal(qe, qe2) and th.equal(qi, qi2)


def test_input_flag_words_and_widths_for_every_flag_combination():
    """Host logic of the _build_inputs flag sets (homophily_controller.py:137-184, 186-201): for all 128 combinations the controller's
    flag words (the rollout heads' input_flags: None with obs_others_last_action; the learner kernel's input_flags_all) and the C
    ABI's layout arithmetic (ssd_build_inputs_width: host code, no launch) agree with the reference's _get_input_shape; FastPolicy
    accepts exactly the sets whose inputs plus the inc head's one-hot action fit the 64-column weight image."""
    import itertools
    from types import SimpleNamespace
    from homophily_marl_amd import abi
    from homophily_marl_amd.controllers import REGISTRY as mac_REGISTRY
    from homophily_marl_amd.fast_policy import FastPolicy
    z, meta = load_fixture("learner_cleanup5.npz")
    args, batch, _, _ = build(z, meta)
    lib = abi.load_library()
    names = ["obs_last_action", "obs_agent_id", "obs_reward", "obs_inc_reward", "obs_distance", "obs_agent_pos", "obs_others_last_action"]
    bits = [1, 2, 4, 8, 16, 32, 64]
    n, A = args.n_agents, args.n_actions
    widths = [A, n, 1, 1, n, 2, n * A]
    shipped = 0
    for combo in itertools.product([False, True], repeat=7):
        a = SimpleNamespace(**dict(vars(args), **dict(zip(names, combo))))
        mac = mac_REGISTRY[a.mac](batch.scheme, {"agents": n}, a)
        word = sum(b for b, on in zip(bits, combo) if on)
        tail = sum(w for w, on in zip(widths, combo) if on)
        assert mac.input_shape == a.obs_dim_net + tail
        assert mac.input_flags_all == word and mac.input_flags == (None if combo[6] else word)
        assert lib.ssd_build_inputs_width(n, A, abi.INPUT_EXPLICIT | word) == tail
        fits = (not combo[6]) and a.obs_dim_net + tail + A <= 64
        assert FastPolicy.supports(mac) == (fits or mac.shipped_flags)
        shipped += mac.shipped_flags
    assert shipped == 1 and lib.ssd_build_inputs_width(n, A, 0) == A + n + 4          # 0 = the shipped set


def test_fused_logs_mapping_reads_the_loss_kernels_sums():
    """The nine logged scalars of a train step (homophily_learner.py:228-246) as a read-only mapping over the loss kernel's column sums:
    the quotients are formed when read, from the CURRENT contents of the buffers (a replayed graph refreshes them in place)."""
    from homophily_marl_amd.learners.homophily_learner import _FusedLogs
    sums = th.arange(13, dtype=th.float32) + 1.0
    dens = th.tensor([4.0, 9.0])
    logs = _FusedLogs(sums, dens, rows=10.0, n=5)
    assert len(logs) == 9 and set(logs) == set(_FusedLogs.KEYS)
    assert float(logs["loss_value_env"]) == 3.0 / 4.0 and float(logs["loss_value_inc"]) == 4.0 / 4.0 and float(logs["loss_sim"]) == 5.0 / 10.0
    assert abs(float(logs["q_env_taken_mean"]) - 6.0 / 10.0) < 1e-7 and abs(float(logs["q_inc_taken_mean"]) - 7.0 / 50.0) < 1e-7
    assert abs(float(logs["value_give_mean"]) - 8.0 / 10.0) < 1e-7 and abs(float(logs["value_receive_mean"]) - 9.0 / 10.0) < 1e-7
    assert abs(float(logs["incentives_to_cleanup_per"]) - 10.0 / (11.0 + 1e-6)) < 1e-6
    sums.mul_(2.0)                                                     # the buffers change (next replay): the mapping follows
    assert float(logs["loss_value_env"]) == 6.0 / 4.0 and dict(logs.items()).keys() == set(_FusedLogs.KEYS)
    with pytest.raises(KeyError):
        logs["nope"]
