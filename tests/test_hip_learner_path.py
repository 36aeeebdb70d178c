"""GPU suite: the learner-side HIP kernels (through the C ABI), the controller / learner on the device, the drop-in env
surface and the vectorised runner."""
import ctypes as C

import numpy as np
import pytest
import torch as th
import torch.nn.functional as F

from homophily_marl_amd import abi

pytestmark = pytest.mark.gpu


def test_build_inputs_kernel_vs_oracle_and_torch():
    from homophily_marl_amd import ops
    from oracle.oracle_py import lib as oracle_lib
    g = th.Generator().manual_seed(1)
    for (B, n, A) in ((1, 3, 9), (257, 5, 9), (64, 10, 8)):
        la = th.randint(0, A, (B, n), generator=g)
        lr = th.randint(-2, 3, (B, n), generator=g).float()
        li = th.randint(0, 3, (B, n, n), generator=g)
        pos = th.randint(1, 40, (B, n, 2), generator=g).float()
        width = A + n + 4
        for t0 in (True, False):
            ref = np.full((B * n, width + 3), -7, np.float32)
            o = oracle_lib()
            o.ssd_cpu_build_inputs(B, n, A, int(t0), la.numpy().ctypes.data, lr.numpy().ctypes.data, li.numpy().ctypes.data,
                                   pos.numpy().ctypes.data, C.c_float(30.8), ref.ctypes.data, width + 3, 2)
            dev = th.full((B * n, width + 3), -7.0, device="cuda")
            ops.build_inputs_tail(dev, 2, la.cuda(), lr.cuda(), li.cuda(), pos.cuda(), 30.8, A, t0)
            cpu = th.full((B * n, width + 3), -7.0)
            ops.build_inputs_tail(cpu, 2, la, lr, li, pos, 30.8, A, t0)
            assert (dev.cpu().numpy() == ref).all(), (B, n, A, t0)
            assert (cpu.numpy() == ref).all(), (B, n, A, t0)


def test_incentive_transfer_kernel_vs_oracle_and_torch():
    from homophily_marl_amd import ops
    from oracle.oracle_py import lib as oracle_lib
    g = th.Generator().manual_seed(2)
    for (B, T, n) in ((1, 2, 3), (16, 101, 5), (7, 13, 10)):
        a = th.randint(0, 3, (B, T, n, n), generator=g)
        r = th.randint(-3, 4, (B, T - 1, n), generator=g).float()
        outs = [np.zeros((B, t, n), np.float32) for t in (T - 1, T, T, T, T - 1, T - 1)]
        oracle_lib().ssd_cpu_incentive_transfer(B, T, n, a.numpy().ctypes.data, r.numpy().ctypes.data, C.c_float(1.0), C.c_float(0.1),
                                                C.c_float(1.0), C.c_float(float(T)), *[x.ctypes.data for x in outs])
        dev = ops.incentive_transfer(a.cuda(), r.cuda(), 1.0, 0.1, 1.0, float(T))
        cpu = ops.incentive_transfer(a, r, 1.0, 0.1, 1.0, float(T))
        for i in range(6):
            assert (dev[i].cpu().numpy() == outs[i]).all(), (B, T, n, i)
            assert (cpu[i].numpy() == outs[i]).all(), (B, T, n, i)


LOG_KEYS = ("loss_value_env", "loss_value_inc", "loss_sim", "value_give_mean", "value_receive_mean", "q_env_taken_mean", "q_inc_taken_mean",
            "incentives_to_cleanup_per", "incentives_to_harvest_per")


@pytest.mark.parametrize("train_graph,storage", [(False, "f32"), (True, "f32"), (False, "code"), (True, "code")])
@pytest.mark.parametrize("name", ["learner_cleanup5.npz", "learner_harvest5.npz", "learner_cleanup5_w4.npz"])
def test_learner_on_device_matches_reference_fixture(name, train_graph, storage):
    """Same fixtures as the CPU suite, network + HIP kernels on the GPU (fused loss kernel, sequence GRU kernels): inputs, Q-values,
    every logged value of two optimisation steps within 1e-5 (fp32) and every parameter after each step (pins the two-Adam /
    double-clip order on the device) -- eagerly and with the train step captured as hipGraphs (the captured step is compared with
    the REFERENCE's numbers, not only with the eager step).  storage "code": the sampled batch holds the observations as u8 class
    codes (obs_storage: code) and both networks' encoders run through the matrix-core encoder kernel (ops.encode_codes: forward
    in the kernel, backward from the activations it emits) -- against the same reference numbers.  `_w4`: the reference run at
    4 x the initial weights (|q| up to 2, the magnitudes of a trained network) -- the split products hold the same absolute bar.
    With code storage every operator must stay on the HIP kernels (strict_device_ops)."""
    from homophily_marl_amd import ops
    from tests.learner_util import build, load_fixture, param_checksums
    th.backends.cuda.matmul.allow_tf32 = False
    z, meta = load_fixture(name)
    args, batch, mac, learner = build(z, meta, device="cuda:0", overrides=dict(train_graph=train_graph), code_obs=storage == "code")
    ops.set_strict(storage == "code")
    try:
        _learner_fixture_body(z, args, batch, mac, learner, train_graph, storage)
    finally:
        ops.set_strict(False)


def _learner_fixture_body(z, args, batch, mac, learner, train_graph, storage):
    from tests.learner_util import param_checksums
    assert learner._fused(batch) and learner.use_graph == train_graph
    assert (batch["obs"].dtype == th.uint8) == (storage == "code")
    with th.no_grad():
        got = mac._build_inputs(batch, 3).cpu().numpy()
        nf = args.obs_dim_net
        assert (got[:, nf:] == z["inputs_t3"][:, nf:]).all()
        q_env, q_inc = learner.unroll(mac, batch)
        assert np.abs(q_env.cpu().numpy() - z["q_env"]).max() < 1e-5
        assert np.abs(q_inc.cpu().numpy() - z["q_inc"]).max() < 1e-5
    if train_graph:
        # the graph path runs its first two calls eagerly and captures at the third: bring a SECOND learner past its capture on a
        # scratch copy of the weights, then rewind weights / optimiser state and replay the two fixture steps through the graphs
        sd0 = {k: v.clone() for k, v in mac.agent.state_dict().items()}
        for _ in range(3):
            learner.train(batch, 0, 0)
        assert learner._graph is not None
        mac.agent.load_state_dict(sd0); learner.target_mac.load_state(mac)
        for opt in (learner.optimiser_env, learner.optimiser_inc):
            for st in opt.state.values():
                st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    for step in range(2):
        if train_graph:
            learner.train(batch, 0, 0)
            logs = learner._static_logs
        else:
            logs = learner.cal_loss_and_step(batch)
        for k in LOG_KEYS:
            assert abs(float(logs[k]) - float(z["step%d_%s" % (step, k)])) < 1e-5, (step, k, float(logs[k]), float(z["step%d_%s" % (step, k)]))
        sums, sqs, heads = param_checksums(mac)
        assert np.abs(heads - z["step%d_param_head" % step]).max() < 2e-5, step
        assert np.abs(sqs - z["step%d_param_sq" % step]).max() / np.abs(z["step%d_param_sq" % step]).max() < 1e-5, step


# the stated bar of the labelled bf16 learner variant: single bf16 products carry 8 significand bits (2^-9 relative per operand)
BF16_Q_TOL = 2e-2           # |q - reference q| absolute, q of order 1 (measured: 2e-4 .. 6e-3 on the three fixtures)
BF16_LOSS_RTOL = 5e-3       # |loss - reference loss| / max(|reference loss|, 1e-2) for the two TD losses and the similarity loss (measured: < 9e-4)


@pytest.mark.parametrize("train_graph", [False, True])
@pytest.mark.parametrize("name", ["learner_cleanup5.npz", "learner_harvest5.npz", "learner_cleanup5_w4.npz"])
def test_bf16_learner_variant_is_close_to_the_reference_and_labelled(name, train_graph):
    """learner_dtype: bf16 (the labelled reduced-precision variant of BASELINE config 2's "homophily Q-learner bf16"; never the
    headline): the per-agent affine layers, the recurrence and the encoder evaluate single bf16 MFMA products (ssd_set_learner_precision
    1), parameters / Adam / loss stay f32.  Against the REFERENCE's numbers of the three learner fixtures: Q-values within BF16_Q_TOL,
    the three losses of two optimisation steps within BF16_LOSS_RTOL relative (the deviation is printed), the parameters after the
    steps close to the reference's (Adam normalises the step size: a sign flip of a tiny gradient moves a weight by 2 lr) -- and NOT
    equal to the fp32 path's (the variant really runs: the same checks at fp32 tolerances must fail).  The fp32 path is restored."""
    from homophily_marl_amd import ops
    from tests.learner_util import build, load_fixture, param_checksums
    z, meta = load_fixture(name)
    args, batch, mac, learner = build(z, meta, device="cuda:0", overrides=dict(train_graph=train_graph, learner_dtype="bf16"), code_obs=True)
    assert learner.precision == 1
    ops.set_strict(True)
    try:
        ops.set_learner_precision(1)
        with th.no_grad():
            q_env, q_inc = learner.unroll(mac, batch)
        de, di = np.abs(q_env.cpu().numpy() - z["q_env"]).max(), np.abs(q_inc.cpu().numpy() - z["q_inc"]).max()
        print("bf16 learner: max |q_env - ref| %.2e  |q_inc - ref| %.2e  (|q| max %.2f)" % (de, di, np.abs(z["q_env"]).max()))
        assert 1e-5 < max(de, di) < BF16_Q_TOL                              # close, and not the f32 path
        if train_graph:
            sd0 = {k: v.clone() for k, v in mac.agent.state_dict().items()}
            for _ in range(3):
                learner.train(batch, 0, 0)
            assert learner._graph is not None
            mac.agent.load_state_dict(sd0); learner.target_mac.load_state(mac)
            for opt in (learner.optimiser_env, learner.optimiser_inc):
                for st in opt.state.values():
                    st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        worst = 0.0
        for step in range(2):
            if train_graph:
                learner.train(batch, 0, 0)
                logs = learner._static_logs
            else:
                ops.set_learner_precision(1)
                logs = learner.cal_loss_and_step(batch)
            for k in ("loss_value_env", "loss_value_inc", "loss_sim"):
                ref = float(z["step%d_%s" % (step, k)])
                rel = abs(float(logs[k]) - ref) / max(abs(ref), 1e-2)
                worst = max(worst, rel)
                assert rel < BF16_LOSS_RTOL, (step, k, float(logs[k]), ref)
            for k in ("value_give_mean", "value_receive_mean"):          # integer counts: exact in any precision
                assert abs(float(logs[k]) - float(z["step%d_%s" % (step, k)])) < 1e-5
        print("bf16 learner: worst relative loss deviation over two steps %.2e" % worst)
        _, sqs, _ = param_checksums(mac)
        assert np.abs(sqs - z["step1_param_sq"]).max() / np.abs(z["step1_param_sq"]).max() < 2e-2
        assert all(bool(th.isfinite(p).all()) for p in mac.parameters())
    finally:
        ops.set_strict(False)
        ops.set_learner_precision(2)
    assert abi.load_library().ssd_learner_precision() == 2


def _random_learner_batch(B, T, n, kind, seed):
    """A synthetic sampled batch with everything the loss reads switched on: rewards of both signs, cleaning, incentives of all three
    kinds, early termination (mask), random availability."""
    from types import SimpleNamespace
    from homophily_marl_amd.components.episode_buffer import EpisodeBatch
    from homophily_marl_amd.components.transforms import OneHot
    from homophily_marl_amd.controllers import REGISTRY as mac_REGISTRY
    from homophily_marl_amd.learners import REGISTRY as le_REGISTRY
    from homophily_marl_amd.run import load_config
    g = th.Generator().manual_seed(seed)
    A = 9 if kind == "cleanup" else 8
    mp = "default10" if (kind == "harvest" or n == 10) else "default5"
    cfg = load_config(kind, overrides=dict(env_args=dict(num_agents=n, map=mp, episode_limit=T, view_size=7), use_cuda=True, batch_size=B))
    args = SimpleNamespace(**cfg)
    args.device, args.n_agents, args.n_actions = "cuda:0", n, A
    args.obs_shape, args.obs_dims = (3, 15, 15), (15, 15)
    args.state_dims = {"default5": (25, 18), "default10": (48, 18) if kind == "cleanup" else (9, 38)}[mp]
    scheme = {"obs": {"vshape": (3, 15, 15), "group": "agents"}, "actions": {"vshape": (1,), "group": "agents", "dtype": th.long},
              "avail_actions": {"vshape": (A,), "group": "agents", "dtype": th.int}, "reward": {"vshape": (n,)},
              "terminated": {"vshape": (1,), "dtype": th.uint8}, "clean_num": {"vshape": (n,)}, "apple_den": {"vshape": (n,)},
              "agent_pos": {"vshape": (n, 2)}, "agent_orientation": {"vshape": (n, 2)},
              "actions_inc": {"vshape": (n, 1), "group": "agents", "dtype": th.long}}
    groups = {"agents": n}
    batch = EpisodeBatch(scheme, groups, B, T + 1, preprocess={"actions": ("actions_onehot", [OneHot(out_dim=A)])}, device="cuda:0")
    avail = (th.rand(B, T + 1, n, A, generator=g) < 0.7).int()
    avail[..., 4] = 1                                                       # STAY is always available
    acts = th.multinomial(avail.reshape(-1, A).float(), 1).reshape(B, T + 1, n, 1)
    ainc = th.randint(0, 3, (B, T + 1, n, n), generator=g) * (1 - th.eye(n, dtype=th.long))
    term = th.zeros(B, T + 1, 1, dtype=th.uint8)
    for b in range(B):
        term[b, T - 1 - (b % 3) * 2] = 1                                    # some episodes end early: the TD mask is not all ones
    obs = th.zeros(B, T + 1, n, 3, 15, 15)
    cls = th.randint(0, 4, (B, T + 1, n, 15, 15), generator=g)
    for ch, c in ((0, 2), (1, 1), (2, 3)):
        obs[:, :, :, ch] = (cls == c).float() * (255.0 / 256.0)
    batch.update(dict(obs=obs, actions=acts, avail_actions=avail, actions_inc=ainc.unsqueeze(-1), terminated=term,
                      reward=th.randint(-1, 3, (B, T + 1, n), generator=g).float() * (th.rand(B, T + 1, n, generator=g) < 0.3),
                      clean_num=th.randint(0, 3, (B, T + 1, n), generator=g).float() * (th.rand(B, T + 1, n, generator=g) < 0.3),
                      apple_den=th.rand(B, T + 1, n, generator=g), agent_pos=th.randint(1, 17, (B, T + 1, n, 2), generator=g).float(),
                      agent_orientation=th.tensor([-1.0, 0.0]).expand(B, T + 1, n, 2)))
    logger = SimpleNamespace(log_stat=lambda *a, **k: None, console_logger=None)
    out = []
    for fused in (True, False):
        th.manual_seed(seed)
        a2 = SimpleNamespace(**vars(args))
        a2.fused_loss = fused
        mac = mac_REGISTRY[a2.mac](batch.scheme, groups, a2).cuda()
        learner = le_REGISTRY[a2.learner](mac, batch.scheme, logger, a2)
        learner.cuda()
        for p in learner.target_mac.parameters():                           # a target net that differs from the live net
            p.data.add_(0.05 * th.randn(p.shape, generator=th.Generator().manual_seed(seed + p.numel())).cuda())
        out.append(learner)
    return batch, out[0], out[1]


@pytest.mark.parametrize("B,T,n,kind", [(16, 100, 5, "cleanup"), (5, 23, 10, "cleanup"), (7, 31, 5, "harvest")])
def test_fused_loss_kernel_matches_the_tensor_op_loss(B, T, n, kind):
    """ssd_td_sim_loss (one launch: incentive transfer, both double-Q TD losses, the similarity loss and the gradient w.r.t. the
    Q-values) against the tensor-op statement of homophily_learner.py:94-217 + autograd on the same batch and weights: denominators,
    every logged value, and the whole flat parameter gradient."""
    batch, fused, plain = _random_learner_batch(B, T, n, kind, seed=B + T)
    assert fused._fused(batch) and not plain._fused(batch)
    d1, d2 = fused.denominators(batch), plain.denominators(batch)
    assert d1[0] == d2[0] and d1[1] == d2[1] and float(d1[1]) > 0 and float(d1[0]) < B * T * n       # sim mask active, TD mask not all ones
    l1 = fused.forward_backward(batch, d1)
    l2 = plain.forward_backward(batch, d2)
    for k in LOG_KEYS:
        assert abs(float(l1[k]) - float(l2[k])) < 1e-5 * max(1.0, abs(float(l2[k]))), (k, float(l1[k]), float(l2[k]))
    # what the logger reads at a log interval: all scalars through ONE device -> host copy; the quotients are formed on the host (a
    # true f32 division there, a multiplication by the reciprocal in torch's device kernel: the last bit may differ)
    extra = (("clean_num_mean", batch["clean_num"][:, :-1].mean()),)
    host = dict(l1.host_items(extra))
    assert host["clean_num_mean"] == float(extra[0][1])
    assert all(abs(host[k] - float(l1[k])) <= 1e-6 * max(1e-3, abs(float(l1[k]))) for k in LOG_KEYS), host
    g1, g2 = fused._flat_grad, plain._flat_grad
    assert float(g2.abs().max()) > 1e-4
    assert float((g1 - g2).abs().max()) < 2e-6 * max(1.0, float(g2.abs().max())), (float((g1 - g2).abs().max()), float(g2.abs().max()))


def test_replayed_train_graph_equals_the_eager_step_with_other_work_in_between(monkeypatch):
    """Regression: every replay of the captured train step must reproduce an eager evaluation of the same step (same weights, batch,
    denominators) while ANOTHER learner trains eagerly between the replays.  With ATen's multi-block row reductions inside the
    graph (bias gradients of baddbmm / Linear / Conv2d, column sums) the third replay returned another reduction's partial sums
    (MI355X, ROCm 7.2) and long runs diverged; the captured step therefore evaluates those sums as GEMMs (ops.column_sums)."""
    from types import SimpleNamespace
    from homophily_marl_amd.controllers import REGISTRY as mac_REGISTRY
    from homophily_marl_amd.learners import REGISTRY as le_REGISTRY
    batch, eager, _ = _random_learner_batch(16, 100, 5, "cleanup", seed=3)
    a = SimpleNamespace(**vars(eager.args)); a.train_graph = True
    mac = mac_REGISTRY[a.mac](batch.scheme, {"agents": 5}, a).cuda()
    mac.agent.load_state_dict(eager.mac.agent.state_dict())
    graph = le_REGISTRY[a.learner](mac, batch.scheme, SimpleNamespace(log_stat=lambda *x, **k: None, console_logger=None), a)
    graph.cuda()
    monkeypatch.setenv("SSD_GRAPH_CHECK", "1")        # learner._graph_step compares each replay with the eager step (SystemExit on a mismatch)
    g = th.Generator(device="cuda").manual_seed(0)
    for step in range(24):
        r = (th.rand(batch["reward"].shape, generator=g, device="cuda") < 0.05).float()
        batch.data.transition_data["reward"].copy_(r)
        graph.train(batch, 0, step)
        eager.train(batch, 0, step)
    assert graph._graph is not None and graph._check_n >= 20
    assert all(bool(th.isfinite(p).all()) for p in mac.parameters())


def test_env_class_drop_in_surface_single_env():
    """REGISTRY['cleanup'](**env_args) with the reference's kwargs behaves like the reference env object (n_env = 1):
    compared call by call with the CPU oracle on the same counter seed."""
    from homophily_marl_amd.envs import REGISTRY
    from oracle.oracle_py import OracleEnv
    env_args = dict(num_agents=5, render=False, episode_limit=30, is_replay=False, view_size=7, map="default5",
                    extra_args=dict(random_spawn_point=False, random_spawn_rotation=0, disable_rotation_action=True,
                                    disable_fire_action=True, obs_color="simplified"), seed=11)
    env = REGISTRY["cleanup"](**env_args)
    orc = OracleEnv("cleanup", map="default5", num_agents=5, n_env=1, view_size=7, episode_limit=30, rng_mode=abi.RNG_COUNTER, seed=11)
    info = env.get_env_info()
    assert info["state_shape"] == (3, 25, 18) and info["obs_shape"] == (3, 15, 15) and info["n_actions"] == 9
    assert info["n_agents"] == 5 and info["episode_limit"] == 30 and info["state_dims"] == (25, 18) and info["obs_dims"] == (15, 15)
    assert env.get_avail_actions() == [[1, 1, 1, 1, 1, 0, 0, 0, 1]] * 5
    env.reset(); orc.reset()
    rng = np.random.default_rng(0)
    term = False
    t = 0
    while not term:
        obs = env.get_obs()
        ref = orc.observe(want_state=True)
        assert len(obs) == 5 and obs[0].dtype == np.float64 and (np.stack(obs) == ref["obs"][0]).all()
        assert (env.get_state() == ref["state"][0]).all()
        assert (env.get_agent_pos() == ref["pos"][0]).all() and (env.get_agent_orientation() == ref["orient"][0]).all()
        acts = rng.choice([0, 1, 2, 3, 4, 8], 5)
        reward, term, einfo = env.step(acts)
        o = orc.step(acts[None])
        assert reward.dtype == np.float64 and (reward == o["reward"][0]).all()
        assert (einfo["clean_num"] == o["clean_num"][0]).all() and (einfo["apple_den"] == o["apple_den"][0]).all()
        assert term == bool(o["terminated"][0])
        t += 1
    assert t == 30 and "collective_return" in einfo and "equality_metric" in einfo
    with pytest.raises(KeyError):
        env.step([9, 0, 0, 0, 0])
    env.close()


def test_vectorised_runner_and_one_training_iteration():
    """hip_vec runner, 64 envs: batch contents are consistent (stored obs == observation of the stored state
    trajectory re-played on the oracle with the stored actions) and one learner.train runs end to end."""
    from homophily_marl_amd.run import load_config, setup, train_iteration
    from oracle.oracle_py import OracleEnv
    N, T = 64, 12
    cfg = load_config("cleanup", overrides=dict(
        runner="hip_vec", batch_size_run=N, batch_size=8, buffer_size=N, buffer_cpu_only=False, store_state=False,
        env_args=dict(num_agents=5, map="default5", episode_limit=T, seed=5), use_cuda=True, save_model=False))
    ctx = setup(cfg)
    batch = ctx.runner.run(test_mode=False)
    assert batch.batch_size == N and batch.max_seq_length == T + 1
    assert int(batch["filled"].sum().item()) == N * (T + 1)
    assert ctx.runner.t_env == N * T
    orc = OracleEnv("cleanup", map="default5", num_agents=5, n_env=N, view_size=7, episode_limit=T, rng_mode=abi.RNG_COUNTER, seed=5)
    orc.reset()
    acts = batch["actions"].squeeze(-1).cpu().numpy()
    for t in range(T):
        ob = orc.observe()
        assert (batch["obs"][:, t].cpu().numpy() == ob["obs"]).all(), t
        assert (batch["agent_pos"][:, t].cpu().numpy() == ob["pos"]).all()
        o = orc.step(acts[:, t] % 9)
        assert (batch["reward"][:, t].cpu().numpy() == o["reward"]).all()
        assert (batch["clean_num"][:, t].cpu().numpy() == o["clean_num"]).all()
        assert (batch["terminated"][:, t, 0].cpu().numpy() == o["terminated"]).all()
    assert (batch["obs"][:, T].cpu().numpy() == orc.observe()["obs"]).all()
    ai = batch["actions_inc"].squeeze(-1)
    assert (ai.diagonal(dim1=2, dim2=3) == 0).all() and int(ai.max()) <= 2       # no self-incentive
    assert (batch["actions_onehot"].argmax(-1) == batch["actions"].squeeze(-1)).all()
    ep = train_iteration(ctx, 0)
    assert ep == N and ctx.buffer.episodes_in_buffer == N
    ctx.runner.close_env()


def test_graph_runner_equals_generic_runner_under_greedy_actions():
    """hip_graph (one hipGraph replay per timestep, persistent storage) must store exactly what hip_vec stores: with
    test_mode=True the action selection is greedy, hence deterministic given equal weights and env seeds."""
    from homophily_marl_amd.run import load_config, setup
    N, T = 32, 10
    batches = {}
    for runner in ("hip_vec", "hip_graph"):
        th.manual_seed(0)
        cfg = load_config("cleanup", overrides=dict(
            runner=runner, batch_size_run=N, batch_size=8, buffer_size=N, buffer_cpu_only=False, store_state=False,
            env_args=dict(num_agents=5, map="default5", episode_limit=T, seed=9), use_cuda=True, save_model=False, runner_stats=False,
            fast_policy=False))      # same torch controller arithmetic in both runners -> bit-identical batches
        ctx = setup(cfg)
        out = []
        for ep in range(3):                       # episode 1 eager, graph captured at episode 2, replayed in 2 and 3
            b = ctx.runner.run(test_mode=True)
            out.append({k: v.clone() for k, v in b.data.transition_data.items()})
        batches[runner] = out
        ctx.runner.close_env()
    for ep in range(3):
        for k in batches["hip_vec"][ep]:
            assert th.equal(batches["hip_vec"][ep][k], batches["hip_graph"][ep][k]), (ep, k)


def test_graph_captured_train_step_equals_eager():
    from tests.learner_util import build, load_fixture
    th.backends.cuda.matmul.allow_tf32 = False
    z, meta = load_fixture("learner_harvest5.npz")
    finals = []
    for graph in (False, True):
        args, batch, mac, learner = build(z, meta, device="cuda:0")
        if graph:
            # rebuild the learner with the graph flag (needs capturable Adam from construction)
            from homophily_marl_amd.learners import REGISTRY as le_REGISTRY
            from types import SimpleNamespace
            args.train_graph = True
            learner = le_REGISTRY[args.learner](mac, batch.scheme, SimpleNamespace(log_stat=lambda *a, **k: None, console_logger=None), args)
            learner.cuda()
            assert learner.use_graph
        args.learner_log_interval = 10 ** 12
        for i in range(5):                        # calls 1-2 eager, capture at 3, replay 3-5
            learner.train(batch, 0, 0)
        finals.append(th.cat([p.detach().reshape(-1) for p in mac.agent.parameters()]).cpu())
        if graph:
            assert learner._graph is not None
    assert (finals[0] - finals[1]).abs().max() < 1e-5, (finals[0] - finals[1]).abs().max()


def test_stepwise_forward_equals_time_batched_unroll():
    from tests.learner_util import build, load_fixture
    z, meta = load_fixture("learner_cleanup5.npz")
    args, batch, mac, learner = build(z, meta, device="cuda:0")
    with th.no_grad():
        q_env, q_inc = mac.unroll(batch)
        mac.init_hidden(batch.batch_size)
        for t in range(batch.max_seq_length):
            a, b, _ = mac.forward(batch, t)
            assert (a - q_env[:, t]).abs().max() < 1e-5 and (b - q_inc[:, t]).abs().max() < 1e-5, t


@pytest.mark.parametrize("groups,kind,n,view,storage,spg", [(1, "cleanup", 5, 7, "f32", 2), (2, "cleanup", 5, 7, "f32", 2), (1, "harvest", 5, 7, "f32", 2),
                                                              (1, "cleanup", 10, 7, "f32", 2), (1, "harvest", 5, 15, "f32", 2),
                                                              (1, "cleanup", 5, 7, "code", 2), (2, "harvest", 5, 15, "code", 2),
                                                              (1, "harvest", 5, 15, "code", 14), (1, "cleanup", 5, 7, "code", 7),
                                                              (1, "cleanup", 5, 7, "f32", 10)])
def test_fast_graph_runner_stores_a_consistent_batch(groups, kind, n, view, storage, spg):
    """hip_graph + FastPolicy (encoder-fused obs store, store-step kernel): the stored batch must be self-consistent
    with the env dynamics (replayed on the CPU oracle with the stored actions), exactly like the generic runner's."""
    from homophily_marl_amd.run import load_config, setup
    from oracle.oracle_py import OracleEnv
    N, T = 48, 14
    mp = "default10" if (kind == "harvest" or n == 10) else "default5"
    th.manual_seed(0)
    cfg = load_config(kind, overrides=dict(
        runner="hip_graph", batch_size_run=N, batch_size=8, buffer_size=N, buffer_cpu_only=False, store_state=False,
        env_args=dict(num_agents=n, map=mp, episode_limit=T, seed=21, view_size=view), use_cuda=True, save_model=False, runner_stats=False,
        policy_groups=groups, obs_storage=storage, steps_per_graph=spg))
    ctx = setup(cfg)
    # an even number of timesteps per graph (2, 14; 10 -> 7 divides T = 14: odd) with one env group takes the pipelined timestep
    # (env head, env step, [inc head of t + encoder of t + 1]); otherwise the four standalone launches
    pipe = groups == 1 and spg in (2, 14)
    ofmt = abi.OBS_CODE if storage == "code" else abi.OBS_F32            # compact storage: u8 class codes instead of f32 planes
    assert ctx.runner.env.native.V == 2 * view + 1
    orc = OracleEnv(kind, map=mp, num_agents=n, n_env=N, view_size=view, episode_limit=T, rng_mode=abi.RNG_COUNTER, seed=21)
    ok_actions = th.nonzero(ctx.runner.env.avail_actions_batch[0, 0]).squeeze(-1).cpu().numpy()
    for ep in range(3):                                  # eager, captured, replayed
        batch = ctx.runner.run(test_mode=False)
        # 15 x 15 and 31 x 31 windows both take the fused matrix-core encoder (it reads class codes: the env's side buffer under
        # f32 storage, the storage itself under code storage) and the heads file the small fields: 4 launches per timestep
        assert ctx.runner.fast is not None and ctx.runner.fast.fused_enc and ctx.runner.fold_store == (groups == 1) and ctx.runner.pipe == pipe
        assert ep == 0 or ctx.runner._graph is not None
        orc.reset()
        acts = batch["actions"].squeeze(-1).cpu().numpy()
        for t in range(T):
            ob = orc.observe(ofmt)
            assert (batch["obs"][:, t].cpu().numpy() == ob["obs"]).all(), (ep, t)
            assert (batch["agent_pos"][:, t].cpu().numpy() == ob["pos"]).all() and (batch["agent_orientation"][:, t].cpu().numpy() == ob["orient"]).all()
            o = orc.step(acts[:, t])
            for k in ("reward", "clean_num", "apple_den"):
                assert (batch[k][:, t].cpu().numpy() == o[k]).all(), (ep, t, k)
            assert (batch["terminated"][:, t, 0].cpu().numpy() == o["terminated"]).all()
        assert (batch["obs"][:, T].cpu().numpy() == orc.observe(ofmt)["obs"]).all()
        assert np.isin(acts, ok_actions).all()
        ai = batch["actions_inc"].squeeze(-1)
        assert (ai.diagonal(dim1=2, dim2=3) == 0).all() and int(ai.max()) <= 2 and int(ai.min()) >= 0
        assert (batch["actions_onehot"].argmax(-1) == batch["actions"].squeeze(-1)).all() and (batch["actions_onehot"].sum(-1) == 1).all()
        assert int(batch["filled"].sum()) == N * (T + 1)
    ctx.runner.close_env()


@pytest.mark.parametrize("kind,n,view,N,slabs,iters", [("cleanup", 5, 7, 4096, 2, 8), ("harvest", 5, 15, 4096, 2, 8), ("cleanup", 10, 7, 8192, 1, 4)])
def test_the_benched_configuration_at_its_own_size(kind, n, view, N, slabs, iters, monkeypatch):
    """Exactly what bench.py times (BASELINE configs 2 / 3 / 4), at its own size: hip_graph runner, 4096 (8192) envs x 100 timesteps, 10
    timesteps per rollout hipGraph (the pipelined 3-launch timestep), class-code storage, episodes written in place into the replay
    buffer (configs 2 / 3: 8 192 slots = two slabs of 4 096; config 4, Cleanup-10 x 8192 -- the looped head kernels, waves that walk
    three tiles --: one slab, so that four iterations reach every kind of replay; its oracle replay is 4 x the work of config 2's),
    train step captured as hipGraphs, device-side replay sampling and runner statistics, strict_device_ops.
    A slab's episodes: 1st eager, 2nd captures the rollout graph, 3rd captures the episode-edge graphs (opening launches: reset, first
    observation, runner-state fills, weight re-pack, encoder of slot 0; closing launches: slot-T pass + statistics), 4th is PURE
    REPLAYS of all three -- eight iterations (four with one slab) put such an episode into every slab, with the weights stepped by
    learner.train in between (a stale re-pack in the opening graph would change the actions the oracle replays).  EVERY rollout is
    replayed on the CPU oracle with the stored actions (the waste permutation persists across episodes): rewards / clean_num /
    apple_den / terminated of every step, pose and observation every 25 steps and at slot T.  Every captured train step is compared,
    gradient by gradient, with an eager evaluation on the same sampled batch (SSD_GRAPH_CHECK)."""
    from homophily_marl_amd import ops
    from homophily_marl_amd.run import load_config, setup
    from oracle.oracle_py import OracleEnv
    T = 100
    mp = "default10" if (kind == "harvest" or n == 10) else "default5"
    th.manual_seed(0)
    np.random.seed(0)
    monkeypatch.setenv("SSD_GRAPH_CHECK", "1")
    cfg = load_config(kind, overrides=dict(
        runner="hip_graph", train_graph=1, steps_per_graph=10, batch_size_run=N, batch_size=16, buffer_size=slabs * N, obs_storage="code",
        buffer_cpu_only=False, store_state=False, strict_device_ops=True,
        env_args=dict(num_agents=n, map=mp, episode_limit=T, seed=1, view_size=view), use_cuda=True, save_model=False))
    ctx = setup(cfg)
    runner, buf, learner = ctx.runner, ctx.buffer, ctx.learner
    if n == 10:      # config 4 runs the looped instantiations of both heads
        assert abi.policy_head_plan(N, n, False)[2] > 1 and abi.policy_head_plan(N, n, True)[2] > 1
    orc = OracleEnv(kind, map=mp, num_agents=n, n_env=N, view_size=view, episode_limit=T, rng_mode=abi.RNG_COUNTER, seed=1)
    ok_actions = th.nonzero(runner.env.avail_actions_batch[0, 0]).squeeze(-1).cpu().numpy()
    ret_sums, ret_sq = [], []
    replayed_edges = 0
    try:
        for it in range(iters):
            b_before = runner._bundles.get(buf["obs"][(it % slabs) * N:].data_ptr())
            pure = b_before is not None and b_before.begin_graph is not None and b_before.finish_graph is not None and b_before.graph is not None
            batch = runner.run(test_mode=False)
            replayed_edges += int(pure)
            assert runner.pipe and runner.fold_store and runner.fast.fused_enc and runner._replay is buf
            assert batch["obs"].data_ptr() == buf["obs"][(it % slabs) * N:].data_ptr()           # written in place, slab it % slabs
            assert it == 0 or (runner._graph is not None and runner._graph_steps == 10)
            orc.reset()
            acts = batch["actions"].squeeze(-1).cpu().numpy()
            rew, cln, den = (batch[k].cpu().numpy() for k in ("reward", "clean_num", "apple_den"))
            term = batch["terminated"][:, :, 0].cpu().numpy()
            ep_ret = rew[:, :T].astype(np.float64).sum(1)
            ret_sums.append(ep_ret.sum()); ret_sq.append((ep_ret * ep_ret).sum())
            for t in range(T):
                if t % 25 == 0:
                    ob = orc.observe(abi.OBS_CODE)
                    assert (batch["obs"][:, t].cpu().numpy() == ob["obs"]).all(), (it, t)
                    assert (batch["agent_pos"][:, t].cpu().numpy() == ob["pos"]).all() and (batch["agent_orientation"][:, t].cpu().numpy() == ob["orient"]).all()
                o = orc.step(acts[:, t])
                assert (rew[:, t] == o["reward"]).all() and (cln[:, t] == o["clean_num"]).all() and (den[:, t] == o["apple_den"]).all(), (it, t)
                assert (term[:, t] == o["terminated"]).all()
            assert (batch["obs"][:, T].cpu().numpy() == orc.observe(abi.OBS_CODE)["obs"]).all()
            assert np.isin(acts, ok_actions).all() and int(batch["filled"].sum()) == N * (T + 1)
            ai = batch["actions_inc"].squeeze(-1)
            assert (ai.diagonal(dim1=2, dim2=3) == 0).all() and int(ai.max()) <= 2 and int(ai.min()) >= 0
            # slot T holds what the closing pass filed (both heads on the last observation): available env actions, a zero diagonal
            assert np.isin(acts[:, T], ok_actions).all()
            buf.insert_episode_batch(batch)
            assert buf.episodes_in_buffer == min(slabs, it + 1) * N
            sample = buf.sample(16, out=learner.sample_out())
            assert (learner.sample_out() is None) or sample is learner.sample_out()
            learner.train(sample, runner.t_env, ctx.train_steps)          # replays are checked against the eager step inside (SystemExit)
            ctx.train_steps += 1
        assert replayed_edges >= slabs                                    # every slab saw an episode made of replays only
        assert learner._graph is not None and learner._check_n >= 2
        assert all(bool(th.isfinite(p).all()) for p in ctx.mac.parameters())
        assert runner.env.native.poll_error() == 0                        # no slot overrun, no f16 range flag
        # device-side runner statistics (episode_runner.py:121-152): the first rollout was logged and cleared (log clock), the others
        # are still accumulated on the device (by the eager launch and by the closing graph's replays) -- against the stored batches' own sums
        acc = runner._acc_train.cpu().numpy()
        assert runner.train_stats["n_episodes"] == (iters - 1) * N and runner.train_stats["n_returns"] == (iters - 1) * N * n
        assert abs(acc[2] - sum(ret_sums[1:])) < 1e-6 * max(1.0, abs(acc[2])) and abs(acc[3] - sum(ret_sq[1:])) < 1e-6 * max(1.0, acc[3])
    finally:
        ops.set_strict(False)
        runner.close_env()


def test_episode_edge_graphs_equal_the_eager_episode_edges():
    """the episode's opening and closing launches as hipGraph replays (episode_edge_graphs, on by default) against the same
    launches issued one by one.  Two runs from the same seeds, seven episodes in ONE storage with learner.train between episodes (the
    opening graph re-packs the weight images: a stale pack would show in the actions), edge graphs on / off: the stored batches
    (slot T included), the runner state an episode leaves (previous actions / reward / incentive actions, returns, hidden states),
    the packed head images and the device-side statistics must be bit-equal; with the graphs on, episodes 4 .. 7 replay both."""
    from homophily_marl_amd.run import load_config, setup
    N, T, n = 96, 12, 5

    def run(edge):
        cfg = load_config("cleanup", overrides=dict(
            runner="hip_graph", train_graph=0, steps_per_graph=4, batch_size_run=N, batch_size=8, buffer_size=N, obs_storage="code",
            buffer_cpu_only=False, store_state=False, episode_edge_graphs=edge, runner_log_interval=10 ** 12,
            env_args=dict(num_agents=n, map="default5", episode_limit=T, seed=9), use_cuda=True, save_model=False))
        th.manual_seed(0)
        np.random.seed(0)
        ctx = setup(cfg)
        runner = ctx.runner
        batches, replays = [], 0
        for it in range(7):
            b = next(iter(runner._bundles.values())) if runner._bundles else None
            replays += int(b is not None and b.begin_graph is not None and b.finish_graph is not None)
            batch = runner.run(test_mode=False)
            batches.append({k: v.clone() for k, v in batch.data.transition_data.items()})
            ctx.buffer.insert_episode_batch(batch)
            ctx.learner.train(ctx.buffer.sample(8), runner.t_env, ctx.train_steps)
            ctx.train_steps += 1
        assert runner.pipe and runner.fold_store
        state = dict(prev_actions=runner.prev_actions, prev_reward=runner.prev_reward, prev_inc=runner.prev_inc, ep_return=runner.ep_return,
                     h_env=runner.fast.h_env, h_inc=runner.fast.h_inc, img_env=runner.fast.p["img_env"], img_inc=runner.fast.p["img_inc"],
                     t_dev=runner.t_dev, acc=runner._acc_train)
        state = {k: v.clone() for k, v in state.items()}
        assert runner.env.native.poll_error() == 0
        runner.close_env()
        return batches, state, replays

    on, s_on, r_on = run(True)
    off, s_off, r_off = run(False)
    assert r_on >= 3 and r_off == 0
    for it, (x, y) in enumerate(zip(on, off)):
        for k in x:
            assert th.equal(x[k], y[k]), (it, k)
    for k in s_on:
        assert th.equal(s_on[k], s_off[k]), k


def test_host_replay_buffer_with_a_captured_train_step():
    """the shipped default buffer_cpu_only: True with use_cuda and train_graph = 1.  From the 4th learner.train on the
    learner offers its device-resident static batch as `out`; a HOST buffer must not gather into it across devices but hand back a
    host sample that train_iteration moves over, as the reference loop does (run.py:207-208)."""
    from homophily_marl_amd.run import load_config, setup, train_iteration
    N, T = 32, 12
    cfg = load_config("cleanup", overrides=dict(
        runner="hip_graph", train_graph=1, batch_size_run=N, batch_size=8, buffer_size=64, buffer_cpu_only=True, store_state=False,
        obs_storage="code", env_args=dict(num_agents=5, map="default5", episode_limit=T, seed=5), use_cuda=True, save_model=False))
    th.manual_seed(0)
    np.random.seed(0)
    ctx = setup(cfg)
    assert str(ctx.buffer.device) == "cpu"
    ep = 0
    for _ in range(7):
        ep = train_iteration(ctx, ep)
    assert ctx.train_steps == 7 and ctx.learner._graph is not None
    assert all(bool(th.isfinite(p).all()) for p in ctx.mac.parameters())
    ctx.runner.close_env()


def test_captured_train_step_equals_the_eager_step_at_batch_64(monkeypatch):
    """batch_size 64: four 16-row tiles per weight set, so the per-tile bias sums of the recurrence (and every other row reduction of
    the step) are added inside the captured graph -- by k_column_sums, not by an ATen multi-block reduction.
    Every replay is compared with an eager evaluation of the same step while another learner trains eagerly in between."""
    from types import SimpleNamespace
    from homophily_marl_amd import ops
    from homophily_marl_amd.controllers import REGISTRY as mac_REGISTRY
    from homophily_marl_amd.learners import REGISTRY as le_REGISTRY
    batch, eager, _ = _random_learner_batch(64, 40, 5, "cleanup", seed=5)
    a = SimpleNamespace(**vars(eager.args)); a.train_graph = True
    mac = mac_REGISTRY[a.mac](batch.scheme, {"agents": 5}, a).cuda()
    mac.agent.load_state_dict(eager.mac.agent.state_dict())
    graph = le_REGISTRY[a.learner](mac, batch.scheme, SimpleNamespace(log_stat=lambda *x, **k: None, console_logger=None), a)
    graph.cuda()
    monkeypatch.setenv("SSD_GRAPH_CHECK", "1")
    g = th.Generator(device="cuda").manual_seed(0)
    ops.set_strict(True)
    try:
        for step in range(40):
            r = (th.rand(batch["reward"].shape, generator=g, device="cuda") < 0.05).float()
            batch.data.transition_data["reward"].copy_(r)
            graph.train(batch, 0, step)
            eager.train(batch, 0, step)
    finally:
        ops.set_strict(False)
    assert graph._graph is not None and graph._check_n >= 36
    assert all(bool(th.isfinite(p).all()) for p in mac.parameters())


def test_fused_gru_gate_kernels_forward_and_backward():
    """ops.gru_gates on the GPU (one fused forward + one fused backward kernel) against the torch expression + autograd."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(3)
    R, H = 10 * 37, 64
    gi = th.randn(R, 3 * H, generator=g, device="cuda", requires_grad=True)
    gh = th.randn(R, 3 * H, generator=g, device="cuda", requires_grad=True)
    h = th.randn(R, H, generator=g, device="cuda", requires_grad=True)
    w = th.randn(R, H, generator=g, device="cuda")
    out = ops.gru_gates(gi, gh, h)
    (out * w).sum().backward()
    got = [out.detach().clone(), gi.grad.clone(), gh.grad.clone(), h.grad.clone()]
    for t in (gi, gh, h):
        t.grad = None
    r = th.sigmoid(gi[..., :H] + gh[..., :H]); z = th.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    cand = th.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    ref = (1 - z) * cand + z * h
    (ref * w).sum().backward()
    exp = [ref.detach(), gi.grad, gh.grad, h.grad]
    for a, b in zip(got, exp):
        assert (a - b).abs().max() < 2e-6, (a - b).abs().max()


@pytest.mark.parametrize("T,G,B", [(7, 3, 16), (5, 10, 21), (3, 2, 5)])
def test_gru_sequence_kernels_match_the_stepwise_recurrence(T, G, B):
    """ssd_gru_seq_fwd / _bwd (one launch for all T) against a plain torch unroll of the cell (homophily_agent.py:162-165):
    states and the gradients wrt the input projections, W_h and b_h; B = 21, 5: ragged 16-row tiles."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(T * 100 + B)
    gi = (th.randn(T, G, B, 192, generator=g, device="cuda") * 0.7).requires_grad_()
    wh = (th.randn(G, 64, 192, generator=g, device="cuda") * 0.15).requires_grad_()
    bh = (th.randn(G, 1, 192, generator=g, device="cuda") * 0.1).requires_grad_()
    wout = th.randn(G, T, B, 64, generator=g, device="cuda")
    hs = ops.gru_sequence(gi, wh, bh)
    (hs * wout).sum().backward()
    got = [x.grad.clone() for x in (gi, wh, bh)]
    for x in (gi, wh, bh):
        x.grad = None
    h = gi.new_zeros(G, B, 64)
    ref = []
    for t in range(T):
        gh = th.baddbmm(bh, h, wh)
        r = th.sigmoid(gi[t][..., :64] + gh[..., :64]); z = th.sigmoid(gi[t][..., 64:128] + gh[..., 64:128])
        cand = th.tanh(gi[t][..., 128:] + r * gh[..., 128:])
        h = (1 - z) * cand + z * h
        ref.append(h)
    ref = th.stack(ref, dim=1)
    (ref * wout).sum().backward()
    assert (hs - ref).abs().max() < 1e-5
    for a, b, name in zip(got, (gi.grad, wh.grad, bh.grad), ("d_gi", "d_wh", "d_bh")):
        assert (a - b).abs().max() < 2e-5 * max(1.0, b.abs().max().item()), name
    with th.no_grad():       # inference form (no saved gates)
        assert (ops.gru_sequence(gi, wh, bh) - ref).abs().max() < 1e-5


@pytest.mark.parametrize("n,R,I,O", [(5, 1616, 64, 192), (5, 1616, 73, 64), (5, 333, 64, 9), (3, 8080, 80, 3), (5, 50, 64, 1), (2, 17, 5, 20),
                                     (1, 8080, 1014, 32), (5, 8081, 80, 1)])      # the last three long-row shapes take the row-chunked dw (2 launches)
def test_bias_bmm_kernels_match_baddbmm_autograd(n, R, I, O):
    """ssd_bias_bmm_fwd / _bwd (csrc/ssd_bmm.hip: the learner's per-agent affine layers, exact-f32 MFMAs) against th.baddbmm and
    its autograd at the learner's shapes and at ragged ones (rows, inputs, outputs not multiples of 16)."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(R + I + O)
    x = th.randn(n, R, I, generator=g, device="cuda").requires_grad_()
    w = (th.randn(n, I, O, generator=g, device="cuda") * 0.2).requires_grad_()
    b = (th.randn(n, 1, O, generator=g, device="cuda") * 0.1).requires_grad_()
    wout = th.randn(n, R, O, generator=g, device="cuda")
    y = ops.bias_bmm(x, w, b)
    (y * wout).sum().backward()
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    ref = th.baddbmm(b, x, w)
    (ref * wout).sum().backward()
    assert (y - ref).abs().max() < 1e-5 * max(1.0, ref.abs().max().item())
    for a, t, name in zip(got, (x, w, b), ("dx", "dw", "db")):
        assert (a - t.grad).abs().max() < 2e-5 * max(1.0, t.grad.abs().max().item()), (name, (a - t.grad).abs().max().item())
    with th.no_grad():
        assert (ops.bias_bmm(x, w, b) - ref).abs().max() < 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,R,I,O", [(5, 1616, 73, 64), (5, 1616, 82, 64), (3, 8080, 80, 3), (2, 17, 5, 20)])
def test_bias_bmm_with_the_leaky_relu_fused_matches_the_two_operators(n, R, I, O):
    """ssd_bias_bmm_leaky_fwd / _bwd (fc1 + nn.LeakyReLU() of both heads, homophily_agent.py:158,182, as one launch per direction)
    against F.leaky_relu(th.baddbmm(...)) and its autograd; zeros in the pre-activation take the negative slope, as torch does."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(R + I + O)
    x = th.randn(n, R, I, generator=g, device="cuda")
    b = th.randn(n, 1, O, generator=g, device="cuda") * 0.1
    x[:, 0] = 0; b[:, :, 0] = 0                                           # an exact zero pre-activation in every weight set
    x.requires_grad_(); b.requires_grad_()
    w = (th.randn(n, I, O, generator=g, device="cuda") * 0.2).requires_grad_()
    wout = th.randn(n, R, O, generator=g, device="cuda")
    y = ops.bias_bmm(x, w, b, leaky=True)
    (y * wout).sum().backward()
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    ref = th.nn.functional.leaky_relu(th.baddbmm(b, x, w))
    (ref * wout).sum().backward()
    assert (y - ref).abs().max() < 1e-5 * max(1.0, ref.abs().max().item())
    for a, t, name in zip(got, (x, w, b), ("dx", "dw", "db")):
        assert (a - t.grad).abs().max() < 2e-5 * max(1.0, t.grad.abs().max().item()), (name, (a - t.grad).abs().max().item())
    with th.no_grad():
        assert (ops.bias_bmm(x, w, b, leaky=True) - ref).abs().max() < 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,T,n,A", [(16, 101, 5, 9), (3, 7, 10, 9), (5, 4, 5, 8)])
def test_unroll_other_kernel_matches_the_tensor_expression(B, T, n, A):
    """ssd_unroll_other (one launch) against the tensor-op assembly it replaces (F.one_hot + division + concatenation + two permuted
    copies): the incentive head's per-receiver features [T * B, n, A + 7] and the one-hot actions agent-major [n, T * B, A], bit for bit."""
    from homophily_marl_amd import ops
    from homophily_marl_amd.modules.agents.homophily_agent import HomophilyAgent
    g = th.Generator(device="cuda").manual_seed(B + T)
    acts = th.randint(0, A, (B, T, n), generator=g, device="cuda")
    pos = th.randint(0, 25, (B, T, n, 2), generator=g, device="cuda").float()
    orient = th.randint(-1, 2, (B, T, n, 2), generator=g, device="cuda").float()
    rew, cln, den = (th.rand(B, T, n, generator=g, device="cuda") for _ in range(3))
    scale = 30.805843601498726
    other, act_tm = ops.unroll_other(acts, pos, orient, rew, cln, den, scale, A)
    onehot = F.one_hot(acts, num_classes=A)
    # pos / scale as the reference forms it: a true f32 division (CPU torch; torch's device kernel multiplies by 1 / scale instead)
    ref = HomophilyAgent.unroll_other(onehot, (pos.cpu() / scale).cuda(), orient, rew, cln, den, th.float32)
    assert th.equal(other, ref)
    assert th.equal(act_tm, onehot.float().permute(2, 1, 0, 3).reshape(n, T * B, A))


@pytest.mark.parametrize("n,T,B,inner,K,E", [(5, 101, 16, 1, 9, 0), (5, 101, 16, 5, 3, 16), (3, 7, 5, 3, 3, 16), (10, 4, 9, 10, 3, 16), (5, 6, 4, 1, 8, 0), (5, 9, 16, 5, 3, 15)])
def test_dueling_head_matches_the_two_layers_and_the_concatenated_rows(n, T, B, inner, K, E):
    """ops.dueling_head (ssd_bias_bmm[2]_fwd + ssd_dueling_head_fwd; backward ssd_dueling_head_bwd + ssd_bias_bmm2_bwd_w + ssd_bias_bmm_bwd_x)
    against the reference's formulation in tensor ops: separate advantage / value layers (homophily_agent.py:166-170,202-207) on rows
    that are, for the incentive head, the materialised concatenations [h_i | other_j] of every ordered pair (:194-201).  Values and
    all gradients (h, both layers' weights and biases) within f32 round-off of the different summation order."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(n * 100 + T)
    H, TB = 64, T * B
    h = (th.randn(n, TB, H, generator=g, device="cuda") * 0.5).requires_grad_()
    other = th.randn(TB, inner, E, generator=g, device="cuda") if E else None
    wa = (th.randn(n, H + E, K, generator=g, device="cuda") * 0.2).requires_grad_()
    wv = (th.randn(n, H + E, 1, generator=g, device="cuda") * 0.2).requires_grad_()
    ba = (th.randn(n, 1, K, generator=g, device="cuda") * 0.2).requires_grad_()
    bv = (th.randn(n, 1, 1, generator=g, device="cuda") * 0.2).requires_grad_()
    wgt = th.randn(B, T, n, inner, K, generator=g, device="cuda")
    q = ops.dueling_head(h, other, th.cat([wa, wv], dim=2), th.cat([ba, bv], dim=2), B, T, inner)
    x = h if other is None else th.cat([h.unsqueeze(2).expand(n, TB, inner, H), other.unsqueeze(0).expand(n, TB, inner, E)], dim=-1).reshape(n, TB * inner, H + E)
    a, v = th.baddbmm(ba, x, wa), th.baddbmm(bv, x, wv)
    ref = (v + a - a.mean(dim=-1, keepdim=True)).reshape(n, T, B, inner, K).permute(2, 1, 0, 3, 4)
    assert (q.reshape(ref.shape) - ref).abs().max() < 2e-6
    got = th.autograd.grad((q.reshape(ref.shape) * wgt).sum(), [h, wa, wv, ba, bv])
    exp = th.autograd.grad((ref * wgt).sum(), [h, wa, wv, ba, bv])
    for x_, y_ in zip(got, exp):
        assert (x_ - y_).abs().max() <= 2e-5 * max(1.0, float(y_.abs().max())), float((x_ - y_).abs().max())


def test_fill_blocks_and_runner_stats_kernels():
    """ssd_fill_blocks (the runner state an episode opens with, one launch) and ssd_runner_stats (one rollout's statistics added to
    the f64 accumulator, episode_runner.py:121-152) against the tensor statements they replace."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(3)
    a = th.randint(0, 9, (4096, 5), generator=g, device="cuda")
    b = th.randn(4096, 5, generator=g, device="cuda")
    c = th.randint(0, 3, (4096, 5, 5), generator=g, device="cuda")
    d = th.ones(1, dtype=th.long, device="cuda")
    e = th.randn(5, 4096, 64, generator=g, device="cuda")
    guard = [t.clone() for t in (a, b)]
    ops.fill_blocks([(a[1:], 0xFFFFFFFF), (b[:-1], 0), (c, 0), (d, 0), (e, 0)])          # (slices: the neighbours must stay)
    assert bool((a[1:] == -1).all()) and bool((a[0] == guard[0][0]).all()) and bool((b[:-1] == 0).all()) and bool((b[-1] == guard[1][-1]).all())
    assert not bool(c.any()) and int(d) == 0 and not bool(e.any())
    coll, eq = th.randn(4096, generator=g, device="cuda") * 30, th.rand(4096, generator=g, device="cuda")
    ret = th.randn(4096, 5, generator=g, device="cuda") * 8
    acc = th.tensor([1.0, 2.0, 3.0, 4.0], dtype=th.float64, device="cuda")
    ops.runner_stats(coll, eq, ret, acc)
    r = ret.double()
    ref = th.tensor([1.0, 2.0, 3.0, 4.0], dtype=th.float64, device="cuda") + th.stack([coll.double().sum(), eq.double().sum(), r.sum(), (r * r).sum()])
    assert float((acc - ref).abs().max()) < 1e-9 * float(ref.abs().max())


def test_graph_runner_with_non_shipped_input_flags():
    """obs_others_last_action / obs_distance switch the controller to the torch input assembly: the graph runner then takes the
    generic (non-FastPolicy) timestep, still captured as a hipGraph, and a train step runs on its batch."""
    from homophily_marl_amd.run import load_config, setup, train_iteration
    N, T, n = 32, 10, 5
    th.manual_seed(0)
    cfg = load_config("cleanup", overrides=dict(
        runner="hip_graph", batch_size_run=N, batch_size=8, buffer_size=N, buffer_cpu_only=False, store_state=False,
        obs_others_last_action=True, obs_distance=True,
        env_args=dict(num_agents=n, map="default5", episode_limit=T, seed=5), use_cuda=True, save_model=False, runner_stats=False))
    ctx = setup(cfg)
    assert not ctx.mac.shipped_flags and ctx.mac.input_shape == 32 + 9 + n + 2 + 9 * n + n + 2
    for ep in range(3):
        batch = ctx.runner.run(test_mode=False)
        assert ctx.runner.fast is None and (ep == 0 or ctx.runner._graph is not None)
        acts = batch["actions"].squeeze(-1)
        assert bool(th.isin(acts, th.tensor([0, 1, 2, 3, 4, 8], device=acts.device)).all())
        assert int(batch["filled"].sum()) == N * (T + 1)
    train_iteration(ctx, 0)
    ctx.runner.close_env()


@pytest.mark.parametrize("shape", [(1616, 64), (5, 1616, 192), (10, 333, 10), (8080, 1014), (1, 7, 3)])
def test_column_sums_kernel(shape):
    """ssd_column_sums (the learner's bias gradients) against a float64 row sum; bit-identical between two launches."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(sum(shape))
    x = th.randn(*shape, generator=g, device="cuda")
    a, b = ops.column_sums(x), ops.column_sums(x)
    ref = x.double().sum(-2)
    assert a.shape == ref.shape and th.equal(a, b)
    assert float((a.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


def test_graph_runner_takes_the_fused_heads_for_flag_sets_that_fit():
    """A non-shipped _build_inputs flag set without obs_others_last_action (here + obs_distance, - obs_reward) stays on the
    FastPolicy kernels (ssd_policy_head.input_flags).  Greedy episodes: every stored env action must be the argmax of the
    Q-values the torch controller computes from the stored batch (mac.unroll = the learner's view of the same inputs) wherever
    the top two are not a numerical tie, and a train step runs on the batch."""
    from homophily_marl_amd.run import load_config, setup, train_iteration
    N, T, n = 48, 12, 5
    th.manual_seed(0)
    cfg = load_config("cleanup", overrides=dict(
        runner="hip_graph", batch_size_run=N, batch_size=8, buffer_size=N, buffer_cpu_only=False, store_state=False,
        obs_distance=True, obs_reward=False,
        env_args=dict(num_agents=n, map="default5", episode_limit=T, seed=5), use_cuda=True, save_model=False, runner_stats=False))
    ctx = setup(cfg)
    assert not ctx.mac.shipped_flags and ctx.mac.input_shape == 32 + 9 + n + 1 + n + 2
    avail = ctx.runner.env.avail_actions_batch[0, 0]
    for ep in range(3):                                  # eager, captured, replayed
        batch = ctx.runner.run(test_mode=True)
        assert ctx.runner.fast is not None and ctx.runner.fast.fused and (ep == 0 or ctx.runner._graph is not None)
        with th.no_grad():
            q_env, _ = ctx.mac.unroll(batch)
        q = q_env[:, :T].masked_fill(avail.view(1, 1, 1, -1) == 0, -float("inf"))
        top2 = q.topk(2, dim=-1).values
        clear = (top2[..., 0] - top2[..., 1]) > 1e-4
        acts = batch["actions"][:, :T].squeeze(-1)
        assert clear.float().mean() > 0.9 and (acts == q.argmax(-1))[clear].all(), ep
    train_iteration(ctx, 0)
    ctx.runner.close_env()


def test_cat_groups_op_forward_and_backward_match_th_cat():
    """ops.cat_groups (ssd_copy_blocks: the GRU parameter images of both heads in one launch, their gradients split in one) against
    th.cat and its autograd; one output left unused (no gradient arrives for it)."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(4)
    mk = lambda *sh: th.randn(*sh, generator=g, device="cuda").requires_grad_()
    groups = [[mk(5, 64, 64) for _ in range(3)], [mk(5, 64, 64) for _ in range(3)], [mk(5, 1, 64) for _ in range(3)],
              [mk(5, 1, 64) for _ in range(3)], [mk(5, 73, 64), mk(5, 73, 7), mk(5, 73, 1)], [mk(3, 2, 5), mk(3, 2, 9)]]
    outs = ops.cat_groups(groups)
    refs = [th.cat(gr, dim=-1) for gr in groups]
    assert all(th.equal(a, b) for a, b in zip(outs, refs))
    ws = [th.randn(r.shape, generator=g, device="cuda") for r in refs]
    flat = [t for gr in groups for t in gr]
    use = [0, 1, 2, 3, 4]                                                        # the last group's output stays unused
    got = th.autograd.grad(sum((outs[i] * ws[i]).sum() for i in use), flat[:15])
    ref = th.autograd.grad(sum((refs[i] * ws[i]).sum() for i in use), flat[:15])
    assert all(a.is_contiguous() and th.equal(a, b) for a, b in zip(got, ref))
    got = th.autograd.grad((ops.cat_groups(groups)[5] * ws[5]).sum(), flat[15:])
    assert all(th.equal(a, b) for a, b in zip(got, th.autograd.grad((th.cat(groups[5], dim=-1) * ws[5]).sum(), flat[15:])))
    lib = abi.load_library()
    assert lib.ssd_copy_blocks(None, 1, None) == abi.SSD_ERR_INVALID and lib.ssd_copy_blocks((abi.SsdBlockCopy * 1)(), 33, None) == abi.SSD_ERR_INVALID


def test_build_inputs_flags_kernel_matches_the_reference_controller():
    """ssd_build_inputs_flags (the device-side _build_inputs tail for ANY flag set, incl. obs_others_last_action / obs_distance and
    blocks switched off) against the REFERENCE controller's output on the cleanup fixture batch (tests/golden/inputs_flags.npz,
    oracle/gen_inputs_golden.py): stepwise (_build_inputs at the fixture's timesteps) and time-batched (the learner's unroll agrees
    with stepping forward())."""
    import json
    import os
    from types import SimpleNamespace
    from homophily_marl_amd.controllers import REGISTRY as mac_REGISTRY
    from tests.learner_util import build, load_fixture
    z, meta = load_fixture("learner_cleanup5.npz")
    zi = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs_flags.npz"))
    flag_sets = json.loads(bytes(zi["flag_sets"]).decode())
    args, batch, _, _ = build(z, meta, device="cuda:0")
    lib = abi.load_library()
    for i, flags in enumerate(flag_sets):
        a = SimpleNamespace(**dict(vars(args), **flags))
        mac = mac_REGISTRY[a.mac](batch.scheme, {"agents": a.n_agents}, a)
        mac.cuda()
        assert not mac.shipped_flags
        assert lib.ssd_build_inputs_width(a.n_agents, a.n_actions, abi.INPUT_EXPLICIT | mac.input_flags_all) == mac.input_shape - a.obs_dim_net
        for t in zi["ts"]:
            ref = zi["tail_%d_t%d" % (i, t)]
            with th.no_grad():
                got = mac._build_inputs(batch, int(t))[:, a.obs_dim_net:].cpu().numpy()
            assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-6, (i, int(t), flags)
        with th.no_grad():
            q_env, q_inc = mac.unroll(batch)
            mac.init_hidden(batch.batch_size)
            for t in range(3):
                qe, qi, _ = mac.forward(batch, t)
                assert (qe - q_env[:, t]).abs().max() < 1e-5 and (qi - q_inc[:, t]).abs().max() < 1e-5
    assert lib.ssd_build_inputs_width(5, 9, abi.INPUT_EXPLICIT | 128) == -1


@pytest.mark.parametrize("name", ["learner_cleanup5.npz", "learner_harvest5.npz"])
def test_clip_adam_kernel_equals_the_tensor_op_optimiser_tail(name):
    """ssd_clip_adam_step (both gradient clips + both Adam steps, the encoder in both groups, as two launches) against the tensor-op
    tail it replaces (clip_grad_norm_ x 2, torch.optim.Adam x 2) over four optimisation steps from the same start: parameters, both
    optimisers' moments and step counters.  grad_norm_clip is lowered so that both clips actually scale."""
    from tests.learner_util import build, load_fixture
    th.backends.cuda.matmul.allow_tf32 = False
    z, meta = load_fixture(name)
    runs = []
    for fused_opt in (False, True):
        args, batch, mac, learner = build(z, meta, device="cuda:0", overrides=dict(train_graph=False, fused_optimiser=fused_opt, grad_norm_clip=0.05))
        for _ in range(4):
            learner.cal_loss_and_step(batch)
        assert (learner._opt_plan not in (None, False)) == fused_opt
        runs.append((learner, [p.detach().clone() for p in learner.params]))
    (la, pa), (lb, pb) = runs
    for x, y in zip(pa, pb):
        assert (x - y).abs().max() < 2e-6, float((x - y).abs().max())
    for oa, ob in ((la.optimiser_inc, lb.optimiser_inc), (la.optimiser_env, lb.optimiser_env)):
        for qa, qb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
            sa, sb = oa.state[qa], ob.state[qb]
            assert float(sa["step"]) == float(sb["step"]) == 4.0
            for k in ("exp_avg", "exp_avg_sq"):
                # after the first step the two runs' parameters differ in the last bits, hence their gradients: 1e-4 of the largest moment
                assert (sa[k] - sb[k]).abs().max() <= 1e-4 * max(1e-12, float(sa[k].abs().max())), k


def test_clip_adam_kernel_spreads_a_non_finite_norm_like_clip_grad_norm():
    """one NaN gradient makes clip_grad_norm_'s coefficient NaN, which torch multiplies into EVERY gradient of the group
    (and through the once-clipped encoder into the other group): the fused tail does the same instead of stepping unclipped."""
    from tests.learner_util import build, load_fixture
    z, meta = load_fixture("learner_cleanup5.npz")
    outs = []
    for fused_opt in (False, True):
        args, batch, mac, learner = build(z, meta, device="cuda:0", overrides=dict(train_graph=False, fused_optimiser=fused_opt))
        for _ in range(2):                                              # the first step creates the optimiser state
            learner.cal_loss_and_step(batch)
        dens = learner.denominators(batch)
        learner.forward_backward(batch, dens)
        inc_only = next(p for p in learner.params_inc if all(p is not q for q in learner.params_env))
        inc_only.grad.view(-1)[0] = float("nan")
        learner.clip_and_step()
        assert (learner._opt_plan not in (None, False)) == fused_opt
        outs.append([bool(th.isnan(p).all()) for p in learner.params])
    assert outs[0] == outs[1] and all(outs[1])


@pytest.mark.parametrize("T,n,B,k", [(7, 5, 16, 4), (5, 3, 32, 2), (9, 5, 16, 1), (4, 2, 7, 2)])
def test_gru_sequence_parts_equals_the_time_major_launch(T, n, B, k):
    """ops.gru_sequence_parts (ssd_gru_seq_fwd_parts / _bwd_parts: the projections as k separately allocated set-major tensors
    [n, T * B, 3H], two of them without a gradient like the target net's) against ops.gru_sequence on their time-major concatenation:
    states and gradients bit for bit.  B = 7: the ragged batch falls back to the padded time-major launch."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(T * 100 + B)
    H = 64
    parts = [(th.randn(n, T * B, 3 * H, generator=g, device="cuda") * 0.5).requires_grad_(i < max(1, k // 2)) for i in range(k)]
    wh = (th.randn(k * n, H, 3 * H, generator=g, device="cuda") * 0.1).requires_grad_()
    bh = (th.randn(k * n, 1, 3 * H, generator=g, device="cuda") * 0.1).requires_grad_()
    w = th.randn(k * n, T, B, H, generator=g, device="cuda")
    hs_parts = ops.gru_sequence_parts(parts, T, B, wh, bh)                 # one tensor [n, T, B, H] per projection part
    assert len(hs_parts) == k and all(h.shape == (n, T, B, H) for h in hs_parts)
    hs = th.cat(list(hs_parts), dim=0)
    gi = th.cat([p.reshape(n, T, B, 3 * H) for p in parts], dim=0).transpose(0, 1).contiguous()
    ref = ops.gru_sequence(gi, wh, bh)
    assert th.equal(hs, ref)
    live = [p for p in parts if p.requires_grad]
    got = th.autograd.grad(sum((h * w[i * n:(i + 1) * n]).sum() for i, h in enumerate(hs_parts)), live + [wh, bh])
    exp = th.autograd.grad((ref * w).sum(), live + [wh, bh])
    for a, b in zip(got, exp):
        assert a.is_contiguous() and th.equal(a, b)
    # the learner's form: the recurrence weights as k separately allocated parts too, and only the FIRST parts' states (the live net's)
    # enter the loss -- the backward walks that prefix of the sets only; the other parts' weights get no gradient at all
    kl = max(1, k // 2)
    whs = [wh.detach()[i * n:(i + 1) * n].clone().requires_grad_() for i in range(k)]
    bhs = [bh.detach()[i * n:(i + 1) * n].clone().requires_grad_() for i in range(k)]
    hs2 = ops.gru_sequence_parts(parts, T, B, whs, bhs)
    assert all(th.equal(a, b) for a, b in zip(hs2, hs_parts))
    got2 = th.autograd.grad(sum((h * w[i * n:(i + 1) * n]).sum() for i, h in enumerate(hs2[:kl])), live + whs + bhs, allow_unused=True)
    wm = w.clone(); wm[kl * n:] = 0
    exp2 = th.autograd.grad((ops.gru_sequence(gi, wh, bh) * wm).sum(), live + [wh, bh])
    nl = len(live)
    for i in range(nl):
        assert th.equal(got2[i], exp2[i])
    for i in range(k):
        gw, gb = got2[nl + i], got2[nl + k + i]
        if B % 16 == 0:
            assert (gw is None) == (i >= kl) and (gb is None) == (i >= kl)
        if gw is not None:
            assert th.equal(gw, exp2[nl][i * n:(i + 1) * n]) and th.equal(gb, exp2[nl + 1][i * n:(i + 1) * n])


def test_replay_sample_into_a_batch_is_one_gather_launch_with_the_same_episodes():
    """ReplayBuffer.sample(batch_size, out=...) on device buffers (ssd_gather_rows: every field in one launch, rows of any byte
    length and alignment) draws what the field-by-field indexing draws."""
    from homophily_marl_amd import ops
    from homophily_marl_amd.components.episode_buffer import EpisodeBatch, ReplayBuffer
    scheme = {"obs": {"vshape": (5, 15, 15), "dtype": th.uint8}, "reward": {"vshape": (5,)}, "terminated": {"vshape": (1,), "dtype": th.uint8},
              "actions": {"vshape": (5, 1), "dtype": th.long}, "odd": {"vshape": (3,), "dtype": th.uint8}}
    T, N = 101, 64
    buf = ReplayBuffer(scheme, {}, N, T, device="cuda")
    g = th.Generator(device="cuda").manual_seed(3)
    for k, v in buf.data.transition_data.items():
        v.copy_(th.randint(0, 200, v.shape, generator=g, device="cuda").to(v.dtype))
    buf.buffer_index, buf.episodes_in_buffer = 0, N
    out = EpisodeBatch(scheme, {}, 16, T, device="cuda")
    np.random.seed(11)
    ref = buf.sample(16)
    buf._sample_calls = 0                     # the same draw again (device-side counter generator: seed, call number)
    got = buf.sample(16, out=out)
    assert got is out
    for k in ref.data.transition_data:
        assert th.equal(ref[k], got[k]), k
    ids = th.tensor([3, 3, 63, 0], device="cuda")
    dst = th.zeros(4, T, 3, dtype=th.uint8, device="cuda")
    assert ops.gather_rows([(buf["odd"], dst)], ids) and th.equal(dst, buf["odd"][ids])      # 303-byte rows: unaligned words + a byte tail
    assert not ops.gather_rows([(buf["odd"], dst[:3])], ids)                                  # shapes that do not fit are refused


def test_replay_indices_are_drawn_on_the_device():
    """ssd_sample_ids (ReplayBuffer.sample of a device-resident buffer, episode_buffer.py:240-244): bit for bit the host restatement
    (ops.sample_ids on a host tensor), distinct and in range at the bench's sizes, and sample() neither consumes numpy's generator
    after its seed is taken nor moves anything across the host boundary that depends on the draw."""
    from homophily_marl_amd import abi, ops
    from homophily_marl_amd.components.episode_buffer import ReplayBuffer
    for seed, call, pop, cnt in [(0x1234567890ABCDEF, 0, 8192, 16), (7, 3, 4096, 16), (2 ** 63 + 5, 2 ** 31 + 1, 17, 16), (1, 1, 64, 64),
                                 (5, 9, 5000, 1024), (11, 0, 1, 1)]:
        dev = ops.sample_ids(seed, call, pop, cnt, th.empty(cnt, dtype=th.long, device="cuda")).cpu()
        host = ops.sample_ids(seed, call, pop, cnt, th.empty(cnt, dtype=th.long))
        assert th.equal(dev, host), (seed, call, pop, cnt)
        assert len(set(dev.tolist())) == cnt and 0 <= int(dev.min()) and int(dev.max()) < pop
    lib = abi.load_library()
    for pop, cnt in [(8, 16), (100, 0), (100, 1025)]:
        with pytest.raises(abi.SsdError):
            abi.check(lib, lib.ssd_sample_ids(1, 0, pop, cnt, th.empty(2048, dtype=th.long, device="cuda").data_ptr(), 0))
    scheme = {"reward": {"vshape": (5,)}}
    buf = ReplayBuffer(scheme, {}, 64, 8, device="cuda")
    buf["reward"].copy_(th.arange(64, device="cuda").view(64, 1, 1).expand(64, 8, 5))
    buf.buffer_index, buf.episodes_in_buffer = 0, 64
    np.random.seed(3)
    a = buf.sample(16)
    state = np.random.get_state()[1].copy()
    b = buf.sample(16)
    assert (np.random.get_state()[1] == state).all()                    # no host draw per sample
    ra, rb = a["reward"][:, 0, 0].cpu(), b["reward"][:, 0, 0].cpu()
    assert len(set(ra.tolist())) == 16 and len(set(rb.tolist())) == 16 and not th.equal(ra, rb)
    assert th.equal(ra.long(), ops.sample_ids(buf._sample_seed, 0, 64, 16, th.empty(16, dtype=th.long)))


@pytest.mark.parametrize("n,T,B,inner,K", [(5, 101, 16, 1, 9), (5, 101, 16, 5, 3), (3, 7, 5, 3, 3), (10, 4, 9, 1, 8)])
def test_dueling_q_kernels_match_the_tensor_expression(n, T, B, inner, K):
    """ops.dueling_q (ssd_dueling_q_fwd / _bwd: v + a - mean(a) of the time-batched heads, written in the batch layout the loss reads)
    against the tensor expression of homophily_agent.py:168-170 / 203-207 and its autograd."""
    from homophily_marl_amd import ops
    g = th.Generator(device="cuda").manual_seed(n * 1000 + T)
    a = th.randn(n, T * B * inner, K, generator=g, device="cuda").requires_grad_()
    v = th.randn(n, T * B * inner, 1, generator=g, device="cuda").requires_grad_()
    q = ops.dueling_q(a, v, B, T, inner)
    ref = (v + a - a.mean(dim=-1, keepdim=True)).reshape(n, T, B, inner, K).permute(2, 1, 0, 3, 4)
    ref = ref.reshape(B, T, n, K) if inner == 1 else ref
    assert q.shape == ref.shape and q.is_contiguous() and (q - ref).abs().max() < 1e-6
    w = th.randn(q.shape, generator=g, device="cuda")
    got = th.autograd.grad((q * w).sum(), [a, v])
    exp = th.autograd.grad((ref * w).sum(), [a, v])
    for x, y in zip(got, exp):
        assert x.shape == y.shape and (x - y).abs().max() < 2e-6
