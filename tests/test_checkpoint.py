"""Checkpoint interchange: file names / formats of the reference's save_models / load_models
(src/learners/homophily_learner.py:276-288, src/controllers/homophily_controller.py:118-123, src/run.py:137-164).

  * tests/golden/ckpt_cleanup3/{agent.th,opt_env.th,opt_inc.th} were WRITTEN BY THE REFERENCE (oracle/gen_checkpoint_golden.py: its
    own th.save calls after two optimisation steps); loading them here and replaying the third step on the recorded batch must give
    the reference's third-step losses and parameters.
  * save_models -> fresh learner -> load_models round trips: identical parameters, optimiser state and next-step losses, with the
    train step eager and (GPU) captured as hipGraphs.
"""
import json
import os

import numpy as np
import pytest
import torch as th

from tests.learner_util import GOLDEN, build, param_checksums

CKPT = os.path.join(GOLDEN, "ckpt_cleanup3")


def _fixture():
    z = np.load(os.path.join(CKPT, "ckpt.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _resume_matches_reference(device):
    z, meta = _fixture()
    args, batch, mac, learner = build(z, meta, device=device)
    learner.load_models(CKPT)
    for opt in (learner.optimiser_env, learner.optimiser_inc):
        st = [s for s in opt.state.values() if s]
        assert st and all(float(s["step"]) == 2.0 for s in st)                     # two reference steps were taken before the save
    return z, mac, learner, learner.cal_loss_and_step(batch)


def test_reference_written_checkpoint_resumes_to_the_reference_third_step():
    z, mac, learner, logs = _resume_matches_reference("cpu")
    assert abs(logs["loss_value_env"].item() - float(z["step2_loss_value_env"])) < 1e-6
    assert abs(logs["loss_value_inc"].item() - float(z["step2_loss_value_inc"])) < 1e-6
    sums, sqs, _ = param_checksums(mac)
    assert list(mac.agent.state_dict().keys()) == [str(k) for k in z["param_names"]]
    np.testing.assert_allclose(sums, z["step2_param_sum"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(sqs, z["step2_param_sq"], rtol=1e-5, atol=1e-6)


def _round_trip(device, tmp_path, train_graph):
    z, meta = _fixture()
    over = dict(train_graph=train_graph)
    args, batch, mac_a, a = build(z, meta, device=device, overrides=over)
    th.manual_seed(1)
    n_before = 4 if train_graph else 2       # the graph path captures at its third call
    for _ in range(n_before):
        a.train(batch, 0, 0)
    a.save_models(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["agent.th", "opt_env.th", "opt_inc.th"]   # the reference's three files
    args, batch_b, mac_b, b = build(z, meta, device=device, overrides=over)
    if train_graph:                          # a learner that already holds captured graphs must survive the load as well
        for _ in range(3):
            b.train(batch_b, 0, 0)
    b.load_models(str(tmp_path))
    for (ka, va), (kb, vb) in zip(mac_a.agent.state_dict().items(), mac_b.agent.state_dict().items()):
        assert ka == kb and th.equal(va, vb), ka
    for tb in b.target_mac.agent.state_dict().values():
        assert tb.device.type == th.device(device).type
    for oa, ob in ((a.optimiser_env, b.optimiser_env), (a.optimiser_inc, b.optimiser_inc)):
        for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
            sa, sb = oa.state[pa], ob.state[pb]
            assert float(sa["step"]) == float(sb["step"]) == n_before
            assert th.equal(sa["exp_avg"], sb["exp_avg"]) and th.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert ob.param_groups[0]["capturable"] == b.use_graph
    # the next steps of the original and of the resumed learner agree (the target nets differ by design: a resume loads the
    # target from agent.th, homophily_learner.py:281-288 -- so align the original's target the same way)
    a.target_mac.load_state(a.mac)
    for _ in range(3 if train_graph else 1):
        a.train(batch, 0, 0)
        b.train(batch_b, 0, 0)
    for (k, va), vb in zip(mac_a.agent.state_dict().items(), mac_b.agent.state_dict().values()):
        assert th.allclose(va, vb, rtol=0, atol=1e-6), k


def test_save_load_round_trip_cpu(tmp_path):
    _round_trip("cpu", tmp_path, False)


@pytest.mark.gpu
@pytest.mark.parametrize("train_graph", [False, True])
def test_save_load_round_trip_on_device(tmp_path, train_graph):
    _round_trip("cuda:0", tmp_path, train_graph)


@pytest.mark.gpu
@pytest.mark.parametrize("train_graph", [False, True])
def test_reference_written_checkpoint_resumes_on_device(train_graph):
    """The reference's optimiser files carry capturable = False and host-side step counters: a graph-capturing learner must
    re-arm its optimisers after the load (and re-capture), an eager one must just continue."""
    z, meta = _fixture()
    args, batch, mac, learner = build(z, meta, device="cuda:0", overrides=dict(train_graph=train_graph))
    if train_graph:
        for _ in range(3):
            learner.train(batch, 0, 0)       # graphs captured on the random-init weights
    learner.load_models(CKPT)
    learner.train(batch, 0, 0)               # the third step of the checkpointed run (eager also on the graph path: re-captured later)
    sums, sqs, _ = param_checksums(mac)
    np.testing.assert_allclose(sums, z["step2_param_sum"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(sqs, z["step2_param_sq"], rtol=2e-5, atol=1e-6)
    for _ in range(4):                       # keeps running through the re-capture
        learner.train(batch, 0, 0)
    assert all(th.isfinite(p).all() for p in mac.parameters())
