"""The exact-fp32 MFMA GEMM behind every linear layer of the aggregation path (``tgmx_sgemm_nt``): its three kernels -- the few-row,
latency-shaped one (M <= 2048, 16 < K <= 512), the many-row one that stages whole lines through LDS (M >= 6144) and the general K-split one -- against a float64 product, on the shapes the TGAT / TGN / TGCN paths run
and on the edges of the dispatch (ragged M / N / K, unaligned views, untrusted row padding, bias + relu, batches)."""

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


@pytest.mark.parametrize('M', [6143, 6144, 8118, 12600, 12601])
@pytest.mark.parametrize('N,K', [(172, 172), (102, 273), (300, 316), (800, 100), (103, 173), (200, 116), (51, 18)])
def test_sgemm_nt_many_rows_against_float64(M, N, K):
    """The shapes of the training step and of cfg 3 (12 600 / 8 118 rows) through the LDS-staged kernel, and both sides of its row
    threshold (6 143 rows: the K-split kernel).  Rows are padded to multiples of 4 floats like the
    library's own operands, and the padding holds NaNs: a kernel that multiplied it would show."""
    from tgm_amd.nn import _ops

    pad = lambda x: (x + 3) // 4 * 4
    g = torch.Generator(device='cpu').manual_seed(M * 1000003 + N * 1009 + K)
    A = torch.full((M, pad(K)), float('nan')).cuda()
    B = torch.full((N, pad(K)), float('nan')).cuda()
    A[:, :K] = torch.randn(M, K, generator=g).cuda()
    B[:, :K] = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    for b, relu in ((None, False), (bias, True)):
        full = torch.full((M, pad(N) + 4), 7.0, device='cuda')
        out = full[:, :N]
        _ops.sgemm_nt(A[:, :K], B[:, :K], out, bias=b, relu=relu)
        ref = _ref(A[:, :K], B[:, :K], b, relu)
        err = (out.double() - ref).abs().max().item()
        assert err <= 2e-5 * (K**0.5), (M, N, K, err)
        assert torch.equal(full[:, N:], torch.full((M, pad(N) + 4 - N), 7.0, device='cuda'))  # nothing past N is written
        out2 = torch.empty_like(full)[:, :N]
        _ops.sgemm_nt(A[:, :K], B[:, :K], out2, bias=b, relu=relu)
        assert torch.equal(out, out2)  # the same launch twice: the same bits


def test_sgemm_nt_many_rows_batches():
    """W_V per head at the training step's size: 12 600 rows, two heads side by side in a row, strided operands (LDS-staged)."""
    from tgm_amd.nn import _ops

    g = torch.Generator(device='cpu').manual_seed(17)
    H, M, N, K, Kp, Np = 2, 12600, 51, 273, 276, 52
    A = torch.randn(M, H * Kp, generator=g).cuda()
    B = torch.randn(H * Np, Kp, generator=g).cuda()
    bias = torch.randn(H, N, generator=g).cuda()
    out = torch.zeros(M, H * Np, device='cuda')
    _ops.sgemm_nt(A, B, out, bias=bias, M=M, N=N, K=K, batch=H, sA=Kp, sB=Np * Kp, sC=Np)
    for h in range(H):
        ref = _ref(A[:, h * Kp:h * Kp + K], B[h * Np:h * Np + N, :K], bias[h], False)
        assert (out[:, h * Np:h * Np + N].double() - ref).abs().max().item() <= 4e-4
        assert torch.equal(out[:, h * Np + N:(h + 1) * Np], torch.zeros(M, Np - N, device='cuda'))


def test_sgemm_nt_many_rows_unaligned_views_equal_the_aligned_call():
    """The kernel a call takes depends on M alone: column-sliced (not 16-byte-aligned) views of the same numbers go through the same
    LDS-staged kernel float by float and give the same bits as the aligned copies."""
    from tgm_amd.nn import _ops

    g = torch.Generator(device='cpu').manual_seed(23)
    M, N, K = 12600, 172, 273
    big_a = torch.randn(M, 301, generator=g).cuda()
    big_b = torch.randn(N, 303, generator=g).cuda()
    A, B = big_a[:, 1:1 + K], big_b[:, 3:3 + K]  # odd column offsets: no 16-byte loads
    o_view = torch.empty(M, N, device='cuda')
    _ops.sgemm_nt(A, B, o_view)
    Ac = torch.zeros(M, 276, device='cuda'); Ac[:, :K] = A
    Bc = torch.zeros(N, 276, device='cuda'); Bc[:, :K] = B
    o_copy = torch.empty(M, N, device='cuda')
    _ops.sgemm_nt(Ac[:, :K], Bc[:, :K], o_copy)
    assert torch.equal(o_view, o_copy)
    assert (o_view.double() - _ref(A, B, None, False)).abs().max().item() <= 4e-4


_PAIR_SCRIPT = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
from tgm_amd.nn import GraphAttentionEmbedding, Time2Vec
out = []
for U, E in ((3000, 5000), (5000, 13000), (8000, 13000)):  # K-split | K-split, K-split | LDS-staged, LDS-staged | LDS-staged
    torch.manual_seed(U)
    enc = GraphAttentionEmbedding(100, 100, 16, Time2Vec(100)).to('cuda').eval()
    x = torch.randn(U, 100, device='cuda')
    lu = torch.randint(1_000_000, 2_000_000, (U,), device='cuda')
    ei = torch.stack([torch.randint(0, U, (E,), device='cuda'), torch.randint(0, U, (E,), device='cuda')])
    t = torch.randint(0, 1_000_000, (E,), device='cuda')
    msg = torch.rand(E, 16, device='cuda')
    with torch.no_grad():
        z = enc(x, lu, ei, t, msg)
    assert torch.isfinite(z).all()
    out.append(hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest())
print('HASHES', ' '.join(out))
'''


def test_paired_gemm_launches_equal_their_solo_launches_bit_for_bit():
    """TransformerConv's projections run as ONE launch of two GEMMs (node projections beside the edge projection).  Whatever kernels the
    two problems take alone -- both K-split, both LDS-staged (M >= 6144), or one of each (cfg 3 early in a stream: ~5 k unique nodes beside
    ~13 k edges) -- the pair gives each problem's own bits: TGMX_GEMM_PAIR=0 (two launches) against the default, in fresh processes
    (the knob is read once)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for knob in ('0', '1'):
        env = dict(os.environ, TGMX_GEMM_PAIR=knob)
        r = subprocess.run([sys.executable, '-c', _PAIR_SCRIPT % root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[knob] = [ln for ln in r.stdout.splitlines() if ln.startswith('HASHES')][-1]
    assert got['0'] == got['1'], got
