"""Host-tensor branches of the operators the vectorised loop calls (CPU suite: no HIP device here).  The same statements are what the
GPU tests check the kernels against."""
import torch as th

from homophily_marl_amd import ops
from homophily_marl_amd.learners.homophily_learner import _FusedLogs


def test_fill_blocks_and_runner_stats_on_host_tensors():
    a, b, t = th.randint(0, 9, (7, 5)), th.randn(7, 5), th.ones(1, dtype=th.long)
    ops.fill_blocks([(a, 0xFFFFFFFF), (b, 0), (t, 0)])
    assert bool((a == -1).all()) and not bool(b.any()) and int(t) == 0
    coll, eq, ret = th.randn(11) * 30, th.rand(11), th.randn(11, 5) * 8
    acc = th.tensor([1.0, 2.0, 3.0, 4.0], dtype=th.float64)
    ops.runner_stats(coll, eq, ret, acc)
    r = ret.double()
    ref = th.tensor([1.0, 2.0, 3.0, 4.0], dtype=th.float64) + th.stack([coll.double().sum(), eq.double().sum(), r.sum(), (r * r).sum()])
    assert th.equal(acc, ref)


def test_fused_logs_read_through_one_copy_give_the_mapping_s_values():
    """_FusedLogs.host_items (what the learner's log interval reads: one device -> host copy) == the key-by-key mapping."""
    g = th.Generator().manual_seed(0)
    sums, dens = th.rand(13, generator=g) * 50, th.tensor([7000.0, 1234.0])
    logs = _FusedLogs(sums, dens, 8000.0, 5)
    extra = (("clean_num_mean", th.tensor(0.25)), ("apple_den_mean", th.tensor(0.5)))
    got = logs.host_items(extra)
    assert [k for k, _ in got] == ["clean_num_mean", "apple_den_mean"] + list(_FusedLogs.KEYS)
    assert got[0][1] == 0.25 and got[1][1] == 0.5
    for k, v in got[2:]:
        assert v == float(logs[k]), k


def test_bias_bmm_host_branch_with_and_without_the_activation():
    g = th.Generator().manual_seed(1)
    x, w, b = th.randn(3, 17, 5, generator=g), th.randn(3, 5, 4, generator=g), th.randn(3, 1, 4, generator=g)
    assert th.equal(ops.bias_bmm(x, w, b), th.baddbmm(b, x, w))
    assert th.equal(ops.bias_bmm(x, w, b, leaky=True), th.nn.functional.leaky_relu(th.baddbmm(b, x, w)))
