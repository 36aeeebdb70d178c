"""Scheme transforms.  Interface of the reference's src/components/transforms.py:4-22 (Transform.transform /
infer_output_info, OneHot(out_dim))."""
import torch as th


class Transform:
    def transform(self, tensor):
        raise NotImplementedError

    def infer_output_info(self, vshape_in, dtype_in):
        raise NotImplementedError


class OneHot(Transform):
    """actions -> actions_onehot (run.py:117-119): float one-hot over the last dimension."""

    def __init__(self, out_dim):
        self.out_dim = out_dim

    def transform(self, tensor):
        idx = tensor.long()
        out = th.zeros(*idx.shape[:-1], self.out_dim, dtype=th.float32, device=tensor.device)
        return out.scatter_(-1, idx, 1.0)

    def infer_output_info(self, vshape_in, dtype_in):
        return (self.out_dim,), th.float32
