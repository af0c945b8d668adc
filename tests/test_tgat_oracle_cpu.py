"""The TGAT oracle (oracle/tgat_ref.py) and the folded restatement the kernels implement
(oracle/tgat_fold.py) against the reference's outputs (goldens g5 / g6).

Tolerance (floating point, fp32 throughout): |got - ref| <= 1e-5 * max(1, |ref|) element-wise,
i.e. 1e-5 relative with an absolute floor of 1e-5 for entries smaller than 1.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import tgat_fold, tgat_ref

RTOL = 1e-5


def close(got, ref, tag):
    err = (got - ref).abs()
    bound = RTOL * ref.abs().clamp(min=1.0)
    worst = (err / bound).max().item()
    assert worst <= 1.0, f'{tag}: worst error {worst:.2f}x the 1e-5 bound (max abs err {err.max().item():.3e})'


def test_time2vec_matches_reference():
    z = np.load(gu.GOLDEN_DIR + '/g6_time2vec.npz')
    t = torch.from_numpy(z['t'])
    w1 = torch.from_numpy((1 / 10 ** np.linspace(0, 9, 100)).reshape(100, 1)).float()
    close(tgat_ref.time2vec(t, w1, torch.zeros(100)), torch.from_numpy(z['out_default']), 'time2vec default init')
    close(tgat_ref.time2vec(t, torch.from_numpy(z['w2']), torch.from_numpy(z['b2'])), torch.from_numpy(z['out_jitter']), 'time2vec jitter')


@pytest.mark.parametrize('case', gu.ATTN_CASES)
@pytest.mark.parametrize('impl', ['ref', 'fold'])
def test_temporal_attention_matches_reference(case, impl):
    meta, a = gu.load(case)
    T = torch.from_numpy
    p = {k[2:]: T(v) for k, v in a.items() if k.startswith('w_')}
    fn = tgat_ref.temporal_attention if impl == 'ref' else tgat_fold.temporal_attention_folded
    out = fn(p, '', meta['n_heads'], T(a['node_x']), T(a['time_feat']), T(a['edge_feat']), T(a['nbr_node_feat']),
             T(a['nbr_time_feat']), T(a['mask']))  # fmt: skip
    close(out, T(a['out']), f'{case}/{impl}')


@pytest.mark.parametrize('case', gu.TGAT_CASES)
@pytest.mark.parametrize('impl', ['ref', 'fold'])
def test_tgat_forward_matches_reference(case, impl):
    meta, params, inputs, z_ref = gu.tgat_case(case)
    fn = tgat_ref.tgat_forward if impl == 'ref' else tgat_fold.tgat_forward_folded
    z = fn(params, meta['dims']['n_heads'], **inputs)
    assert z.shape == z_ref.shape and z.dtype == torch.float32
    close(z, z_ref, f'{case}/{impl}')


def test_reference_self_noise_fixture_is_consistent():
    """tests/golden/g5_self_noise.json (make_golden.py g5n, build container): the replay that measured the reference against itself
    reproduced every stored g5 output bit for bit, and the reference's own float32-vs-float64 distance -- the scale the GPU parity
    checks print their errors against -- is what the float bar's docstring says it is (pure relative 1e-5 is BELOW it)."""
    import json
    import os

    import golden_util as gu

    with open(os.path.join(gu.GOLDEN_DIR, 'g5_self_noise.json')) as f:
        noise = json.load(f)
    cases = [k for k in noise if not k.startswith('_')]
    assert set(cases) == set(gu.ATTN_CASES) | set(gu.TGAT_CASES)
    for k in cases:
        assert noise[k]['reproduces_fixture'], k
        n = noise[k]['float32_vs_float64']
        assert 0 < n['max_abs'] < 1e-5 and n['worst_multiple_of_bound'] < 1.0, (k, n)  # the hybrid bound holds for the reference itself
    assert noise['g5_tgat_example_dims']['float32_vs_float64']['worst_relative_where_ref_ge_1e_2'] > 1e-5
