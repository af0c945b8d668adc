"""ORACLE (test infrastructure only -- never imported by the product path).

The counter-based dropout masks of the kernels (csrc/common.h: counter_u32 / dropout_scale;
include/tgm_amd.h: tgmx_dropout_t) restated in numpy, so that the CPU oracles can run the
reference's train-mode arithmetic (tgm/nn/modules/attention.py:119,126: ``nn.Dropout`` on the
attention weights and on the W_O output) with EXACTLY the masks the device used.  The reference
draws its masks from torch's generator; which elements are dropped is not part of its contract --
the arithmetic given a mask is, and that is what the parity tests pin.
"""
from __future__ import annotations

import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def counter_u32(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """32 uniform bits per index: the splitmix64 finaliser over seed ^ stream * G ^ idx * C (uint64 wrap-around)."""
    with np.errstate(over='ignore'):
        i = idx.astype(np.uint64)
        x = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(stream & 0xFFFFFFFFFFFFFFFF) * np.uint64(0x9E3779B97F4A7C15)) ^ (i * np.uint64(0xD1B54A32D192ED03))
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(32)).astype(np.uint32)


def dropout_scale(p: float, seed: int, stream: int, shape, row0: int = 0) -> torch.Tensor:
    """float32 tensor of ``shape`` [rows, ...]: 1 / (1 - p) where the element is kept, 0 where it is dropped.
    Element index = (row0 + row) * prod(shape[1:]) + offset inside the row, as in the kernels."""
    shape = tuple(int(s) for s in shape)
    if not p:
        return torch.ones(shape, dtype=torch.float32)
    n = int(np.prod(shape))
    width = n // shape[0] if shape[0] else 1
    idx = np.arange(n, dtype=np.uint64) + np.uint64(row0 * width)
    thresh = min(int(float(np.float32(p)) * 4294967296.0), 4294967295)
    keep = counter_u32(seed, stream, idx) >= np.uint32(thresh)
    inv = np.float32(1.0 / (1.0 - float(np.float32(p))))
    return torch.from_numpy(np.where(keep, inv, np.float32(0)).astype(np.float32).reshape(shape))
