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
