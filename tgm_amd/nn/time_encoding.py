"""``Time2Vec`` -- cos(Linear(1 -> T)(t)) with the fixed geometric frequency init
(tgm/nn/modules/time_encoding.py:6-24; same parameter names: ``w.weight``, ``w.bias``)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import _ops


class Time2Vec(nn.Module):
    def __init__(self, time_dim: int) -> None:
        super().__init__()
        self.time_dim = time_dim
        self.w = nn.Linear(1, time_dim)
        freq = (1 / 10 ** np.linspace(0, 9, time_dim)).reshape(time_dim, 1)
        self.w.weight = nn.Parameter(torch.from_numpy(freq).float())
        self.w.bias = nn.Parameter(torch.zeros(time_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [...] int64 or float timestamps / deltas -> [..., time_dim] (HIP kernel, forward only)."""
        return _ops.time2vec(x, self.w.weight.detach().reshape(-1), self.w.bias.detach())
