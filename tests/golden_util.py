"""Helpers shared by the parity tests: load the golden fixtures and replay a
loader schedule (batches, epochs, resets, train/val segments)."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterator, List, Tuple

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name: str) -> Tuple[dict, Dict[str, np.ndarray]]:
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    arrays = {k: z[k] for k in z.files if k != 'meta'}
    meta = json.loads(bytes(z['meta']).decode()) if 'meta' in z.files else {}
    return meta, arrays


def sampler_cases(prefixes=('g1_', 'g2_', 'g4_')) -> List[str]:
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))
    return [n for n in names if n.startswith(tuple(prefixes))]


def schedule(meta: dict) -> Iterator[Tuple[str, int, int]]:
    """Yield ('reset', 0, 0) or ('batch', lo, hi) in the order the fixture was recorded."""
    bs = meta['batch_size']
    for step in meta['plan']:
        if step == 'reset':
            yield 'reset', 0, 0
            continue
        for lo, hi in meta['segments']:
            for s in range(lo, hi, bs):
                yield 'batch', s, min(s + bs, hi)


def batch_starts(meta: dict) -> List[int]:
    bs = meta['batch_size']
    out: List[int] = []
    for lo, hi in meta['segments']:
        out += list(range(lo, hi, bs))
    return out


def seeds_for(meta: dict, a: Dict[str, np.ndarray], lo: int, hi: int):
    parts_n = [a['src'][lo:hi], a['dst'][lo:hi]]
    parts_t = [a['ts'][lo:hi], a['ts'][lo:hi]]
    if meta['has_neg']:
        parts_n.append(a['neg'][lo:hi])
        parts_t.append(a['ts'][lo:hi])
    return np.concatenate(parts_n).astype(np.int32), np.concatenate(parts_t).astype(np.int64)
