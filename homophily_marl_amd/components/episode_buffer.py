"""Episode storage with the reference's surface (src/components/episode_buffer.py:6-254): a scheme-driven dict of
[batch, time, (agents), *vshape] tensors with a `filled` mask, slicing, the one-hot preprocess hook, and a ring replay
buffer with uniform sampling.  Written for device-resident rollouts: `update` accepts tensors already on the device and
copies them without a host round trip; list / numpy inputs are converted exactly like the reference does."""
from types import SimpleNamespace

import numpy as np
import torch as th

from .. import ops


def _count(idx, size):
    if isinstance(idx, slice):
        return len(range(*idx.indices(size)))
    return len(idx)


class EpisodeBatch:
    def __init__(self, scheme, groups, batch_size, max_seq_length, data=None, preprocess=None, device="cpu"):
        self.scheme = dict(scheme)
        self.groups = groups
        self.batch_size = batch_size
        self.max_seq_length = max_seq_length
        self.preprocess = {} if preprocess is None else preprocess
        self.device = device
        if data is not None:
            self.data = data
        else:
            self.data = SimpleNamespace(transition_data={}, episode_data={})
            self._allocate(self.scheme, groups, batch_size, max_seq_length, self.preprocess)

    # ---- allocation ------------------------------------------------------------------------------------------
    def _allocate(self, scheme, groups, batch_size, max_seq_length, preprocess):
        for key, (new_key, transforms) in (preprocess or {}).items():
            assert key in scheme
            vshape, dtype = self.scheme[key]["vshape"], self.scheme[key].get("dtype", th.float32)
            for tr in transforms:
                vshape, dtype = tr.infer_output_info(vshape, dtype)
            entry = {"vshape": vshape, "dtype": dtype}
            for opt in ("group", "episode_const"):
                if opt in self.scheme[key]:
                    entry[opt] = self.scheme[key][opt]
            self.scheme[new_key] = entry
        assert "filled" not in scheme, '"filled" is a reserved key for masking.'
        scheme.update({"filled": {"vshape": (1,), "dtype": th.long}})
        for key, info in scheme.items():
            assert "vshape" in info, "Scheme must define vshape for {}".format(key)
            vshape = info["vshape"]
            vshape = (vshape,) if isinstance(vshape, int) else tuple(vshape)
            group = info.get("group", None)
            if group:
                assert group in groups, "Group {} must have its number of members defined in _groups_".format(group)
                vshape = (groups[group],) + vshape
            dtype = info.get("dtype", th.float32)
            if info.get("episode_const", False):
                self.data.episode_data[key] = th.zeros((batch_size,) + vshape, dtype=dtype, device=self.device)
            else:
                self.data.transition_data[key] = th.zeros((batch_size, max_seq_length) + vshape, dtype=dtype, device=self.device)

    def extend(self, scheme, groups=None):
        self._allocate(scheme, self.groups if groups is None else groups, self.batch_size, self.max_seq_length, None)

    def to(self, device):
        for store in (self.data.transition_data, self.data.episode_data):
            for k in store:
                store[k] = store[k].to(device)
        self.device = device

    # ---- writes ----------------------------------------------------------------------------------------------
    def update(self, data, bs=slice(None), ts=slice(None), mark_filled=True):
        slices = self._parse_slices((bs, ts))
        for k, v in data.items():
            if k in self.data.transition_data:
                target, sl = self.data.transition_data, tuple(slices)
                if mark_filled:
                    target["filled"][sl] = 1
                    mark_filled = False
            elif k in self.data.episode_data:
                target, sl = self.data.episode_data, slices[0]
            else:
                raise KeyError("{} not found in transition or episode data".format(k))
            dtype = self.scheme[k].get("dtype", th.float32)
            if isinstance(v, th.Tensor):
                v = v.detach().to(device=self.device, dtype=dtype)
            else:
                v = th.tensor(np.asarray(v), dtype=dtype, device=self.device)
            dest = target[k][sl]
            self._check_safe_view(v, dest)
            target[k][sl] = v.reshape(dest.shape)
            if k in self.preprocess:
                new_k, transforms = self.preprocess[k]
                out = target[k][sl]
                for tr in transforms:
                    out = tr.transform(out)
                target[new_k][sl] = out.reshape(target[new_k][sl].shape)

    @staticmethod
    def _check_safe_view(v, dest):
        idx = v.dim() - 1
        for s in dest.shape[::-1]:
            if idx < 0 or v.shape[idx] != s:
                if s != 1:
                    raise ValueError("Unsafe reshape of {} to {}".format(tuple(v.shape), tuple(dest.shape)))
            else:
                idx -= 1

    # ---- reads -----------------------------------------------------------------------------------------------
    def __getitem__(self, item):
        if isinstance(item, str):
            if item in self.data.episode_data:
                return self.data.episode_data[item]
            if item in self.data.transition_data:
                return self.data.transition_data[item]
            raise ValueError(item)
        if isinstance(item, tuple) and all(isinstance(it, str) for it in item):
            nd = SimpleNamespace(transition_data={}, episode_data={})
            for key in item:
                if key in self.data.transition_data:
                    nd.transition_data[key] = self.data.transition_data[key]
                elif key in self.data.episode_data:
                    nd.episode_data[key] = self.data.episode_data[key]
                else:
                    raise KeyError("Unrecognised key {}".format(key))
            scheme = {key: self.scheme[key] for key in item}
            groups = {self.scheme[key]["group"]: self.groups[self.scheme[key]["group"]] for key in item if "group" in self.scheme[key]}
            return EpisodeBatch(scheme, groups, self.batch_size, self.max_seq_length, data=nd, device=self.device)
        sl = self._parse_slices(item)
        nd = SimpleNamespace(transition_data={}, episode_data={})
        for k, v in self.data.transition_data.items():
            nd.transition_data[k] = v[tuple(sl)]
        for k, v in self.data.episode_data.items():
            nd.episode_data[k] = v[sl[0]]
        return EpisodeBatch(self.scheme, self.groups, _count(sl[0], self.batch_size), _count(sl[1], self.max_seq_length),
                            data=nd, device=self.device)

    def _parse_slices(self, items):
        if isinstance(items, (slice, int, list, np.ndarray, th.Tensor)):
            items = (items, slice(None))
        if isinstance(items[1], list):
            raise IndexError("Indexing across Time must be contiguous")
        parsed = []
        for it in items:
            if isinstance(it, (list, np.ndarray)):      # one host -> device transfer of the indices, not one per field
                it = th.as_tensor(np.asarray(it), dtype=th.long, device=self.device)
            parsed.append(slice(it, it + 1) if isinstance(it, int) else it)
        return parsed

    def max_t_filled(self):
        return th.sum(self.data.transition_data["filled"], 1).max(0)[0]

    def __repr__(self):
        return "EpisodeBatch. Batch Size:{} Max_seq_len:{} Keys:{} Groups:{}".format(
            self.batch_size, self.max_seq_length, self.scheme.keys(), self.groups.keys())


class ReplayBuffer(EpisodeBatch):
    def __init__(self, scheme, groups, buffer_size, max_seq_length, preprocess=None, device="cpu"):
        super().__init__(scheme, groups, buffer_size, max_seq_length, preprocess=preprocess, device=device)
        self.buffer_size = buffer_size
        self.buffer_index = 0
        self.episodes_in_buffer = 0
        # device-resident buffers draw their sample indices on the device (ops.sample_ids: counter generator keyed by a seed taken
        # from numpy's global generator at the first device sample -- np.random.seed still makes a run reproducible -- and the
        # number of the sample call); host buffers keep the reference's np.random.choice and consume nothing else
        self._sample_seed = None
        self._sample_calls = 0
        self._sample_ids = None

    def reserve(self, n):
        """A view over the next n episode slots, for producers that write whole episodes in place (vectorised runners): hand
        the filled view to insert_episode_batch, which then only advances the ring indices instead of copying.  None if the
        slots would wrap around."""
        if self.buffer_index + n > self.buffer_size:
            return None
        view = self[self.buffer_index:self.buffer_index + n]
        view._reserved = (self, self.buffer_index)
        return view

    def insert_episode_batch(self, ep_batch):
        n = ep_batch.batch_size
        res = getattr(ep_batch, "_reserved", None)
        if res is not None and res[0] is self and res[1] == self.buffer_index:      # written in place (reserve)
            self.buffer_index += n
            self.episodes_in_buffer = max(self.episodes_in_buffer, self.buffer_index)
            self.buffer_index %= self.buffer_size
            return
        if self.buffer_index + n <= self.buffer_size:
            where = slice(self.buffer_index, self.buffer_index + n)
            self.update(ep_batch.data.transition_data, where, slice(0, ep_batch.max_seq_length), mark_filled=False)
            self.update(ep_batch.data.episode_data, where)
            self.buffer_index += n
            self.episodes_in_buffer = max(self.episodes_in_buffer, self.buffer_index)
            self.buffer_index %= self.buffer_size
        else:
            left = self.buffer_size - self.buffer_index
            self.insert_episode_batch(ep_batch[0:left, :])
            self.insert_episode_batch(ep_batch[left:, :])

    def can_sample(self, batch_size):
        return self.episodes_in_buffer >= batch_size

    def sample(self, batch_size, out=None):
        """batch_size episodes drawn without replacement (episode_buffer.py:240-244).  out: an EpisodeBatch of batch_size whole
        episodes (a learner's static copy of the sampled batch) that receives the episodes directly -- one gather per field
        instead of a gather into fresh tensors and a second copy into the consumer's buffers; taken only when it lives on the
        buffer's device.  A device-resident buffer draws the indices on the device (ssd_sample_ids): no host draw, no H2D copy."""
        assert self.can_sample(batch_size)
        if self.episodes_in_buffer == batch_size:
            return self[:batch_size]
        on_device = th.device(self.device).type == "cuda"
        if on_device and batch_size <= ops.abi.SAMPLE_IDS_MAX:
            if self._sample_seed is None:
                self._sample_seed = int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 32)
            if self._sample_ids is None or self._sample_ids.numel() != batch_size:
                self._sample_ids = th.empty(batch_size, dtype=th.long, device=self.device)
            ids = ops.sample_ids(self._sample_seed, self._sample_calls, self.episodes_in_buffer, batch_size, self._sample_ids)
            self._sample_calls += 1
        else:
            ep_ids = np.random.choice(self.episodes_in_buffer, batch_size, replace=False)   # episode_buffer.py:243
            ids = th.as_tensor(ep_ids, dtype=th.long, device=self.device)
        if (out is not None and out.batch_size == batch_size and out.max_seq_length == self.max_seq_length and not self.data.episode_data
                and all(out.data.transition_data[k].device == v.device for k, v in self.data.transition_data.items())):
            pairs = [(v, out.data.transition_data[k]) for k, v in self.data.transition_data.items()]
            if not (ids.is_cuda and ops.gather_rows(pairs, ids)):       # device buffers: one launch for all fields
                ops._leaving_kernels("gather_rows", ids, "fields that do not qualify for ssd_gather_rows")
                for v, dst in pairs:
                    th.index_select(v, 0, ids, out=dst)
            return out
        return self[ids]

    def sample_latest(self, batch_size):
        ep_ids = np.arange(self.buffer_index - batch_size, self.buffer_index) % self.buffer_size
        return self[ep_ids]

    def __repr__(self):
        return "ReplayBuffer. {}/{} episodes. Keys:{} Groups:{}".format(
            self.episodes_in_buffer, self.buffer_size, self.scheme.keys(), self.groups.keys())
