"""ReplayBuffer (reference src/components/episode_buffer.py:207-250) incl. the in-place producer protocol (reserve)."""
import numpy as np
import torch as th

from homophily_marl_amd.components.episode_buffer import EpisodeBatch, ReplayBuffer


def _scheme():
    return {"obs": {"vshape": (3,), "group": "agents"}, "reward": {"vshape": (2,)}, "terminated": {"vshape": (1,), "dtype": th.uint8}}


def _episodes(n, T, seed):
    g = th.Generator().manual_seed(seed)
    b = EpisodeBatch(_scheme(), {"agents": 2}, n, T)
    b.update({"obs": th.randn(n, T, 2, 3, generator=g), "reward": th.randn(n, T, 2, generator=g),
              "terminated": th.zeros(n, T, 1, dtype=th.uint8)}, slice(None), slice(0, T))
    return b


def test_insert_wraps_like_the_reference_ring():
    buf = ReplayBuffer(_scheme(), {"agents": 2}, 10, 4)
    a, b, c = _episodes(4, 4, 1), _episodes(4, 4, 2), _episodes(4, 4, 3)
    for x in (a, b, c):
        buf.insert_episode_batch(x)
    assert buf.buffer_index == 2 and buf.episodes_in_buffer == 10
    assert th.equal(buf["obs"][8:10], c["obs"][0:2]) and th.equal(buf["obs"][0:2], c["obs"][2:4])
    assert th.equal(buf["obs"][4:8], b["obs"])


def test_reserve_is_insert_without_the_copy():
    ref = ReplayBuffer(_scheme(), {"agents": 2}, 8, 4)
    inp = ReplayBuffer(_scheme(), {"agents": 2}, 8, 4)
    for seed in (1, 2, 3):
        ep = _episodes(4, 4, seed)
        ref.insert_episode_batch(ep)
        view = inp.reserve(4)
        assert view is not None and view["obs"].data_ptr() == inp["obs"][inp.buffer_index:].data_ptr()
        for k in ("obs", "reward", "terminated", "filled"):
            view[k].copy_(ep[k])                      # the producer writes in place
        inp.insert_episode_batch(view)
        assert inp.buffer_index == ref.buffer_index and inp.episodes_in_buffer == ref.episodes_in_buffer
    for k in ("obs", "reward", "terminated", "filled"):
        assert th.equal(inp[k], ref[k])
    # a stale reservation (index moved on) falls back to the copying path; a wrapping reservation is refused
    odd = ReplayBuffer(_scheme(), {"agents": 2}, 6, 4)
    v = odd.reserve(4)
    odd.insert_episode_batch(_episodes(4, 4, 9))
    assert odd.reserve(4) is None
    odd.insert_episode_batch(v)
    assert odd.buffer_index == 2 and odd.episodes_in_buffer == 6
    np.random.seed(0)
    assert odd.sample(3).batch_size == 3


def test_sample_into_a_given_batch_draws_the_same_episodes():
    """sample(batch_size, out=...) gathers straight into a consumer's batch (the learner's static copy of the captured train step):
    same np.random draw, same episodes as the plain sample (episode_buffer.py:240-244), no new tensors."""
    buf = ReplayBuffer(_scheme(), {"agents": 2}, 12, 4)
    for seed in (1, 2, 3):
        buf.insert_episode_batch(_episodes(4, 4, seed))
    np.random.seed(7)
    ref = buf.sample(5)
    out = EpisodeBatch(_scheme(), {"agents": 2}, 5, 4)
    ptrs = {k: v.data_ptr() for k, v in out.data.transition_data.items()}
    np.random.seed(7)
    got = buf.sample(5, out=out)
    assert got is out and all(v.data_ptr() == ptrs[k] for k, v in out.data.transition_data.items())
    for k in ref.data.transition_data:
        assert th.equal(ref[k], got[k]), k
    assert buf.sample(5, out=EpisodeBatch(_scheme(), {"agents": 2}, 4, 4)).batch_size == 5      # a batch of another size is not used


def test_sample_with_an_out_batch_on_another_device_falls_back_to_indexing():
    """a host buffer (buffer_cpu_only) asked to sample into a batch that lives elsewhere (the learner's device-resident
    static batch) must not gather across devices: it returns a fresh host batch, as without `out`."""
    buf = ReplayBuffer(_scheme(), {"agents": 2}, 12, 4)
    for seed in (1, 2, 3):
        buf.insert_episode_batch(_episodes(4, 4, seed))
    elsewhere = EpisodeBatch(_scheme(), {"agents": 2}, 5, 4, device="meta")
    np.random.seed(7)
    ref = buf.sample(5)
    np.random.seed(7)
    got = buf.sample(5, out=elsewhere)
    assert got is not elsewhere and got["obs"].device.type == "cpu"
    for k in ref.data.transition_data:
        assert th.equal(ref[k], got[k]), k


def test_sample_ids_restatement_draws_distinct_uniform_ids():
    """ops.sample_ids on host tensors = the arithmetic of ssd_sample_ids (the GPU suite compares the two bit for bit): count distinct
    ids in range, a pure function of (seed, call), every id when count == population, and uniform inclusion frequencies."""
    from homophily_marl_amd import ops
    out = th.empty(16, dtype=th.long)
    a = ops.sample_ids(0x1234567890ABCDEF, 5, 8192, 16, out).clone()
    assert len(set(a.tolist())) == 16 and 0 <= int(a.min()) and int(a.max()) < 8192
    assert th.equal(ops.sample_ids(0x1234567890ABCDEF, 5, 8192, 16, th.empty(16, dtype=th.long)), a)
    assert not th.equal(ops.sample_ids(0x1234567890ABCDEF, 6, 8192, 16, th.empty(16, dtype=th.long)), a)
    assert sorted(ops.sample_ids(3, 0, 16, 16, th.empty(16, dtype=th.long)).tolist()) == list(range(16))
    hits = np.zeros(40)
    first = np.zeros(40)
    for call in range(4000):
        ids = ops.sample_ids(99, call, 40, 8, th.empty(8, dtype=th.long)).numpy()
        hits[ids] += 1
        first[ids[0]] += 1
    assert abs(hits / 4000 - 0.2).max() < 0.03          # inclusion probability 8 / 40 for every id (sd 0.0063)
    assert abs(first / 4000 - 1 / 40).max() < 0.012     # and every id equally likely in a given position (sd 0.0025)
