"""TGCN oracle vs the reference's TGCN class (golden g10; GCNConv = restated third-party op)."""
import pytest
import torch

import golden_util as gu
from oracle.tgcn_ref import tgcn_cell_ref

RTOL = 1e-5


def close(got, ref, tag):
    err = (got - ref).abs()
    worst = (err / (RTOL * ref.abs().clamp(min=1.0))).max().item()
    assert worst <= 1.0, f'{tag}: {worst:.2f}x the 1e-5 bound'


def snapshots(a, tag):
    T = torch.from_numpy
    params = {k[len(tag) + 3 :]: T(v) for k, v in a.items() if k.startswith(f'{tag}_w_')}
    for s in range(3):
        ew = T(a[f'{tag}_s{s}_ew']) if f'{tag}_s{s}_ew' in a else None
        yield params, T(a[f'{tag}_s{s}_x']), T(a[f'{tag}_s{s}_ei']), ew, T(a[f'{tag}_s{s}_H'])


@pytest.mark.parametrize('tag', ['plain', 'improved'])
def test_tgcn_oracle_matches_reference(tag):
    _, a = gu.load('g10_tgcn')
    H = None
    for params, x, ei, ew, H_ref in snapshots(a, tag):
        H = tgcn_cell_ref(params, x, ei, ew, H, improved=tag == 'improved')
        close(H, H_ref, tag)
