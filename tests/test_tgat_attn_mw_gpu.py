"""The four-wave attention kernel (``tgat_attn_reduce_mw_kernel``: one workgroup per row, the slots split over its waves) takes every
launch of at most 2048 rows; larger launches run one wave per row.  A row's arithmetic must not depend on which kernel computed it:
the same rows through ``tgmx_tgat_attn_reduce`` as a launch of R <= 2048 and as the head of a launch of R > 2048 -- equal BIT FOR BIT
(scores summed over the lanes in one fixed order, one fma chain over the slots in ascending order handed from wave to wave)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _inputs(R, k, d, D, T, H, seed, pad_frac=0.45, empty_frac=0.2, big_dt=False):
    g = torch.Generator().manual_seed(seed)
    C = d + D + T
    st = torch.randint(1_000_000, 2_600_000, (R,), generator=g)
    nt = (st[:, None] - torch.randint(1, 900_000, (R, k), generator=g)).clamp(min=0)
    nid = torch.randint(0, 1000, (R, k), generator=g, dtype=torch.int32)
    # right-aligned windows like the sampler's: the first n_pad slots of a row are pads (id -1, time 0)
    n_pad = torch.where(torch.rand(R, generator=g) < empty_frac, torch.full((R,), k), (torch.rand(R, generator=g) * (k + 1) * pad_frac * 2).long().clamp(max=k))
    slot = torch.arange(k)[None, :]
    pad = slot < n_pad[:, None]
    nid[pad] = -1
    nt[pad] = 0
    # a few rows with holes in the middle (arbitrary masks are legal inputs)
    holes = torch.rand(R, k, generator=g) < 0.05
    nid[holes & (torch.arange(R)[:, None] % 7 == 3)] = -1
    if big_dt:  # Time2Vec arguments past the float reduction's range on some rows: the double path
        st[::5] += 1 << 33
    ex = torch.rand(R, k, D, generator=g)
    ex[pad] = 0.0
    nbrf = torch.randn(R, k, d, generator=g)
    qf = torch.randn(R, H, C, generator=g) * 0.1
    w = torch.from_numpy(1 / 10 ** np.linspace(0, 9, T)).float()
    b = torch.randn(T, generator=g) * 0.1
    return [t.to(DEV) for t in (qf, nbrf, ex, st, nt, nid, w, b)]


def _run(lib, native, tensors, R, k, d, D, T, H):
    qf, nbrf, ex, st, nt, nid, w, b = tensors
    zbar = torch.full((R, H, d + D + T), float('nan'), device=DEV)
    native.check(lib.tgmx_tgat_attn_reduce(qf.data_ptr(), nbrf.data_ptr(), d, ex.data_ptr(), D, st.data_ptr(), nt.data_ptr(), nid.data_ptr(), w.data_ptr(),
                                           b.data_ptr(), 0, 0, T, H, k, R, 0.125, 0, zbar.data_ptr(), 0, None, native.stream_ptr()), 'attn_reduce')
    torch.cuda.synchronize()
    return zbar


@pytest.mark.parametrize('k,d,D,T,H', [(20, 1, 172, 100, 2), (20, 172, 172, 100, 2), (10, 8, 12, 16, 2), (15, 4, 16, 100, 2), (5, 1, 4, 8, 1),
                                       (20, 3, 8, 128, 1), (3, 1, 172, 100, 2)])
@pytest.mark.parametrize('big_dt', [False, True])
def test_four_wave_rows_equal_one_wave_rows(k, d, D, T, H, big_dt):
    from tgm_amd import _native

    lib = _native.load()
    R_big, R_small = 2600, 1500
    tensors = _inputs(R_big, k, d, D, T, H, seed=k * 1000 + d * 10 + H, big_dt=big_dt)
    z_one = _run(lib, _native, tensors, R_big, k, d, D, T, H)       # > 2048 rows: a wave per row
    z_four = _run(lib, _native, tensors, R_small, k, d, D, T, H)    # the same first rows, <= 2048: four waves per row
    assert not torch.isnan(z_four).any() and not torch.isnan(z_one).any()
    same = z_one[:R_small] == z_four
    assert bool(same.all()), f'{int((~same).sum())} elements differ, max |d| = {(z_one[:R_small] - z_four).abs().max().item():.3e}'
