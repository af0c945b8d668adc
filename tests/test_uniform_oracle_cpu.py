"""The uniform-sampler restatement against the reference's recorded outputs (golden g11): exact, including the rows
the reference drew with random.sample (same seed, same consumption order)."""
import json
import os
import random

import numpy as np
import pytest

from oracle import uniform_ref

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ['g11_uniform_sparse', 'g11_uniform_dense', 'g11_uniform_dense_directed']


def load(case):
    z = np.load(os.path.join(HERE, 'golden', case + '.npz'))
    meta = json.loads(bytes(z['meta']).decode())
    return meta, z


@pytest.mark.parametrize('case', CASES)
def test_uniform_oracle_matches_reference(case):
    meta, z = load(case)
    src, dst, ts, neg = z['src'], z['dst'], z['ts'], z['neg']
    edge_x = z['edge_x'] if 'edge_x' in z.files else None
    bs, ks = meta['batch_size'], meta['num_nbrs']
    random.seed(meta['random_seed'])
    sampled_rows = 0
    for b in range(meta['num_batches']):
        lo, hi = b * bs, min((b + 1) * bs, len(src))
        seeds = np.concatenate([src[lo:hi], dst[lo:hi], neg[lo:hi]])
        times = np.concatenate([ts[lo:hi]] * 3)
        hops = uniform_ref.step(src, dst, ts, edge_x, seeds, times, ks, int(ts[lo:hi].min()), meta['directed'])
        for h, (sn, st, n, t, x) in enumerate(hops):
            assert np.array_equal(sn, z[f'b{b}_h{h}_seed_nids']), f'b{b} h{h} seeds'
            assert np.array_equal(n, z[f'b{b}_h{h}_nbr_nids']), f'b{b} h{h} ids'
            assert np.array_equal(t, z[f'b{b}_h{h}_nbr_edge_time']), f'b{b} h{h} times'
            assert np.array_equal(x, z[f'b{b}_h{h}_nbr_edge_x']), f'b{b} h{h} feats'
            sampled_rows += int((n[:, -1] >= 0).sum())
    if 'dense' in case:
        assert sampled_rows > 0  # the sampling branch was exercised
