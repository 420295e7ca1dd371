"""Parameter containers of the small MLPs (reference model/mlps.py): DenseBlock (:52-63, FiLM generator) and
DenseResidualBlock (:33-50, the Versa head's hyper-networks).

Same state_dict keys as the reference (`block.0` Linear, `block.1` LayerNorm, `block.3` Linear); the
arithmetic runs inside the grouped FiLM-generator kernel (csrc/film.hip), so these modules are never called.
"""
import torch.nn as nn


class DenseBlock(nn.Module):
    def __init__(self, in_size, hidden_size, out_size):
        super().__init__()
        self.block = nn.Sequential(
            nn.Linear(in_size, hidden_size),
            nn.LayerNorm(hidden_size),
            nn.ReLU(),
            nn.Linear(hidden_size, out_size),
        )

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("DenseBlock is evaluated by orbit_filmgen_forward, not by torch")


class DenseResidualBlock(nn.Module):
    """linear1 -> ELU -> linear2 -> ELU -> linear3 (+ x when the sizes match); evaluated row-wise by orbit_dense_rows."""

    def __init__(self, in_size, out_size):
        super().__init__()
        self.linear1 = nn.Linear(in_size, out_size)
        self.linear2 = nn.Linear(out_size, out_size)
        self.linear3 = nn.Linear(out_size, out_size)
        self.in_size, self.out_size = in_size, out_size

    def forward(self, x):
        """x [R, in] on the HIP device (R <= 16 rows, one per class) -> [R, out]."""
        import torch

        from .. import _lib
        _lib.require_gpu()
        lib, st = _lib.load(), _lib.stream_handle()
        x = x.detach().contiguous().float()
        R = x.shape[0]
        if R > 16:
            raise ValueError("DenseResidualBlock is evaluated for at most 16 rows (classes) at a time")
        ELU, NONE = 3, 0
        h = x
        for i, (lin, act) in enumerate(((self.linear1, ELU), (self.linear2, ELU), (self.linear3, NONE))):
            res = x if (i == 2 and self.in_size == self.out_size) else None
            y = torch.empty(R, lin.out_features, device=x.device, dtype=torch.float32)
            _lib.check(lib.orbit_dense_rows(_lib.dptr(h), R, lin.in_features, _lib.dptr(lin.weight.detach().contiguous()),
                                            _lib.dptr(lin.bias.detach().contiguous()), lin.out_features, act,
                                            _lib.dptr(res), _lib.dptr(y), st), "orbit_dense_rows")
            h = y
        return h
