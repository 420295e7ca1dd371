"""Parameter container of the FiLM generator's MLP (reference model/mlps.py:52-63, DenseBlock).

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
