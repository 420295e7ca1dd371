"""Set encoder (DeepSets task embedding) on the native runtime.

Mirror of reference model/set_encoders.py:34-134: `SetEncoder.forward` encodes every frame of the support
set with SimplePrePoolNet (5 x [conv3x3 -> BN -> ReLU -> maxpool2] -> global average -> 64-d) — here one
native plan whose convolutions run on the MFMA implicit-GEMM kernel with BN, ReLU and the 2x2 max-pool fused
into the epilogue — and `aggregate` averages the per-frame embeddings into the task embedding.
"""
import torch
import torch.nn as nn

from .. import _lib
from .feature_extractors import HipNetwork


class SetEncoder(HipNetwork):
    def __init__(self):
        super().__init__("set_encoder")

    def _leaf_shape(self, key, numel):
        if key.endswith(".0.weight"):
            return (64, numel // (64 * 9), 3, 3)
        return (numel,)

    def forward(self, x, out=None, check_sync=True):
        return super().forward(self._flatten(x), out=out, check_sync=check_sync)

    def _flatten(self, x):
        return x.flatten(end_dim=1) if x.dim() == 5 else x

    def aggregate(self, x, aggregation="mean"):
        if not isinstance(x, torch.Tensor):
            x = torch.cat(x, dim=0)
        if aggregation == "mean":
            _lib.require_gpu()
            if x.requires_grad and torch.is_grad_enabled():
                from .autograd import SetMeanFunction
                return SetMeanFunction.apply(x)
            x = x.contiguous().float()
            out = torch.empty(1, x.shape[1], device=x.device, dtype=torch.float32)
            _lib.check(_lib.load().orbit_set_mean(_lib.dptr(x, torch.float32), x.shape[0], x.shape[1],
                                                  _lib.dptr(out), _lib.stream_handle()), "orbit_set_mean")
            return out
        elif aggregation == "none":
            return x
        raise ValueError(f"Aggregation method {aggregation} not valid!")

    @property
    def output_size(self):
        return 64

    @output_size.setter
    def output_size(self, value):  # HipNetwork.__init__ assigns the native value (64)
        pass


class NullSetEncoder(nn.Module):
    def forward(self, x):
        return None

    def aggregate(self, x, aggregation="mean"):
        return None

    @property
    def output_size(self):
        return None
