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


def tgat_case(name: str):
    """Load a g5_tgat_* fixture -> (meta, params dict, inputs dict of torch tensors, expected z).
    When the fixture does not store nbr_edge_x (example dims: too large), the sampler
    outputs are regenerated with the CPU ring port from the seeded stream and checked
    against the stored ids/times."""
    import torch

    meta, a = load(name)
    T = torch.from_numpy
    params = {k[2:]: T(v) for k, v in a.items() if k.startswith('w_')}
    L = len(meta['num_nbrs'])
    hops = []
    if meta['nbr_edge_x_stored']:
        for h in range(L):
            hops.append({key: T(a[f'h{h}_{key}']) for key in ('seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x')})
    else:
        from oracle.ring_port import RingSamplerCPU
        from tgm_amd.synth import make_stream

        st = make_stream(seed=meta['stream_seed'], **meta['stream'])
        rng = np.random.default_rng(meta['neg_seed'])
        neg = T(rng.integers(meta['neg_lo'], st.num_nodes, st.num_edges).astype(np.int32))
        model = RingSamplerCPU(st.num_nodes, meta['num_nbrs'], st.edge_dim)
        bs = meta['batch_size']
        for b in range(meta['batch_index'] + 1):
            lo, hi = b * bs, min((b + 1) * bs, st.num_edges)
            out = model.step(torch.cat([st.src[lo:hi], st.dst[lo:hi], neg[lo:hi]]), torch.cat([st.ts[lo:hi]] * 3),
                             st.src[lo:hi], st.dst[lo:hi], st.ts[lo:hi], st.edge_x[lo:hi])  # fmt: skip
        for h, (sn, stt, nn, nt, nx) in enumerate(out):
            assert np.array_equal(nn.numpy(), a[f'h{h}_nbr_nids']) and np.array_equal(nt.numpy(), a[f'h{h}_nbr_edge_time'])
            hops.append(dict(seed_nids=sn, seed_times=stt, nbr_nids=nn, nbr_edge_time=nt, nbr_edge_x=nx))
    inputs = dict(
        node_x=T(a['node_x']),
        seed_nids=[h['seed_nids'] for h in hops], seed_times=[h['seed_times'] for h in hops],
        nbr_nids=[h['nbr_nids'] for h in hops], nbr_edge_x=[h['nbr_edge_x'] for h in hops],
        nbr_edge_time=[h['nbr_edge_time'] for h in hops],
    )  # fmt: skip
    return meta, params, inputs, T(a['z'])


TGAT_CASES = ['g5_tgat_small_unix', 'g5_tgat_small_nd8', 'g5_tgat_one_layer', 'g5_tgat_example_dims']
ATTN_CASES = ['g5_attn_pad', 'g5_attn_nopad', 'g5_attn_h4']
