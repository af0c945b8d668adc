"""The exact-fp32 MFMA GEMM behind every linear layer of the aggregation path (``tgmx_sgemm_nt``): both kernels -- the few-row,
latency-shaped one (M <= 2048, 16 < K <= 512) and the general one -- against a float64 product, on the shapes the TGAT / TGN / TGCN
paths run and on the edges of the dispatch (ragged M / N / K, unaligned views, bias + relu, batches)."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(A, B, bias, relu):
    y = A.double() @ B.double().t()
    if bias is not None:
        y = y + bias.double()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize('M', [1, 31, 600, 2048, 2049])
@pytest.mark.parametrize('N,K', [(1, 17), (86, 273), (172, 172), (172, 173), (896, 172), (86, 444), (33, 512), (40, 513), (64, 16), (200, 100)])
def test_sgemm_nt_against_float64(M, N, K):
    from tgm_amd.nn import _ops

    g = torch.Generator(device='cpu').manual_seed(M * 1000003 + N * 1009 + K)
    A = torch.randn(M, K, generator=g).cuda()
    B = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    for b, relu in ((None, False), (bias, True)):
        out = torch.full((M, N), float('nan'), device='cuda')
        _ops.sgemm_nt(A, B, out, bias=b, relu=relu)
        ref = _ref(A, B, b, relu)
        err = (out.double() - ref).abs().max().item()
        assert err <= 2e-5 * (K**0.5), (M, N, K, err)
        # deterministic: the same launch twice gives the same bits
        out2 = torch.empty_like(out)
        _ops.sgemm_nt(A, B, out2, bias=b, relu=relu)
        assert torch.equal(out, out2)


def test_sgemm_nt_unaligned_views_and_padded_output():
    from tgm_amd.nn import _ops

    g = torch.Generator(device='cpu').manual_seed(7)
    big_a = torch.randn(600, 301, generator=g).cuda()
    big_b = torch.randn(172, 303, generator=g).cuda()
    A = big_a[:, 1:274]  # K = 273 from an odd column: no 16-byte loads
    B = big_b[:, 3:276]
    out = torch.zeros(600, 180, device='cuda')
    _ops.sgemm_nt(A, B, out[:, :172], N=172)
    ref = _ref(A, B, None, False)
    assert (out[:, :172].double() - ref).abs().max().item() <= 4e-4
    assert torch.equal(out[:, 172:], torch.zeros(600, 8, device='cuda'))  # nothing past N is written


def test_sgemm_nt_batches():
    from tgm_amd.nn import _ops

    g = torch.Generator(device='cpu').manual_seed(11)
    H, M, N, K = 2, 600, 86, 276
    A = torch.randn(M, H * K, generator=g).cuda()      # heads side by side in a row
    B = torch.randn(H * N, K, generator=g).cuda()      # heads stacked
    out = torch.empty(M, H * N, device='cuda')
    _ops.sgemm_nt(A, B, out, M=M, N=N, K=K, batch=H, sA=K, sB=N * K, sC=N)
    for h in range(H):
        ref = _ref(A[:, h * K:(h + 1) * K], B[h * N:(h + 1) * N], None, False)
        assert (out[:, h * N:(h + 1) * N].double() - ref).abs().max().item() <= 4e-4


def test_small_and_general_kernels_agree_to_rounding(monkeypatch):
    """M = 2048 runs the few-row kernel, M = 2049 the general one: the shared rows differ by summation order only."""
    from tgm_amd.nn import _ops

    g = torch.Generator(device='cpu').manual_seed(13)
    A = torch.randn(2049, 273, generator=g).cuda()
    B = torch.randn(172, 273, generator=g).cuda()
    o1 = torch.empty(2048, 172, device='cuda')
    o2 = torch.empty(2049, 172, device='cuda')
    _ops.sgemm_nt(A[:2048], B, o1)
    _ops.sgemm_nt(A, B, o2)
    assert (o1 - o2[:2048]).abs().max().item() <= 1e-4
