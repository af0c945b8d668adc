"""GPU parity of the discrete-time path: HIP TGCN / GCNConv vs golden g10 and the oracle (1e-5 relative)."""
import pytest
import torch

import golden_util as gu
from test_tgcn_oracle_cpu import close, snapshots

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('tag', ['plain', 'improved'])
def test_tgcn_matches_reference(tag):
    from tgm_amd.nn import TGCN

    meta, a = gu.load('g10_tgcn')
    cell = TGCN(meta['Fin'], meta['C'], improved=tag == 'improved').to(DEV).eval()
    H = None
    for params, x, ei, ew, H_ref in snapshots(a, tag):
        cell.load_state_dict(params)
        H = cell(x.to(DEV), ei.to(DEV), None if ew is None else ew.to(DEV), H)
        close(H.cpu(), H_ref, tag)


def test_tgcn_trade_shaped_vs_oracle():
    """tgbn-trade-like yearly snapshots: 255 nodes, a few thousand weighted edges, embed 128."""
    from oracle.tgcn_ref import gcn_conv_ref, tgcn_cell_ref
    from tgm_amd.nn import GCNConv, TGCN

    torch.manual_seed(1)
    N, Fin, C = 255, 64, 128
    cell = TGCN(Fin, C).to(DEV).eval()
    params = {k: v.detach().cpu() for k, v in cell.state_dict().items()}
    H = H_ref = None
    for snap in range(4):
        E = 3000 + 500 * snap
        ei = torch.randint(0, N, (2, E))
        ew = torch.rand(E) + 0.1
        x = torch.randn(N, Fin)
        H = cell(x.to(DEV), ei.to(DEV), ew.to(DEV), H)
        H_ref = tgcn_cell_ref(params, x, ei, ew, H_ref)
        close(H.cpu(), H_ref, f'snapshot {snap}')
    conv = GCNConv(Fin, C, add_self_loops=False).to(DEV)
    out = conv(x.to(DEV), ei.to(DEV))
    close(out.cpu(), gcn_conv_ref(x, ei, None, conv.lin.weight.detach().cpu(), conv.bias.detach().cpu(), add_self_loops=False), 'gcn no loops')


@pytest.mark.parametrize('N,Fin,C,weighted,improved', [(255, 16, 32, False, False), (255, 64, 128, True, True), (7, 3, 5, True, False), (1000, 16, 32, False, False)])
def test_tgcn_forward_as_one_call_equals_the_composed_sequence(N, Fin, C, weighted, improved, monkeypatch):
    """TGCN.forward in inference = ONE native call (tgmx_tgcn_forward: the 13 launches of a snapshot issued back to back from C, stacked weights
    cached against the parameters' versions, scratch kept between snapshots) -- bit for bit what the Python-composed sequence of the same
    entry points gives (TGMX_TGCN_PY=1; with edge weights: to the last-bit noise of the degree sums' float atomics), over a recurrence of snapshots with changing edge counts, after an in-place parameter update, and for
    a deep copy of the cell (the cached argument block and scratch stay behind)."""
    import copy

    from tgm_amd.nn import TGCN

    torch.manual_seed(N + C)
    cell = TGCN(Fin, C, improved=improved).to(DEV).eval()
    twin = copy.deepcopy(cell)
    H1 = H2 = None
    with torch.no_grad():
        for snap in range(5):
            E = 50 + 700 * snap
            ei = torch.randint(0, N, (2, E), device=DEV)
            if snap % 2:
                ei = ei.int()  # (a DGBatch's endpoints are int32: read as they are by the one-call path)
            ew = (torch.rand(E, device=DEV) + 0.1) if weighted else None
            x = torch.randn(N, Fin, device=DEV)
            monkeypatch.delenv('TGMX_TGCN_PY', raising=False)
            H1 = cell(x, ei, ew, H1)
            monkeypatch.setenv('TGMX_TGCN_PY', '1')
            H2 = twin(x, ei, ew, H2)
            # (weighted edges: A_hat's degree sums are float atomics -- their order, hence the last bit, differs from run to run on EITHER path)
            same = torch.allclose(H1, H2, rtol=1e-5, atol=1e-6) if weighted else torch.equal(H1, H2)
            assert same, f'snapshot {snap}: max |d| = {(H1 - H2).abs().max().item():.3e}'
            H2 = H1.clone()  # (the recurrences continue from one state)
            if snap == 2:  # the cached stacked weights follow the parameters
                for m in (cell, twin):
                    m.conv_r.lin.weight.mul_(1.5)
                    m.linear_c.bias.add_(0.25)
        monkeypatch.delenv('TGMX_TGCN_PY', raising=False)
        clone = copy.deepcopy(cell)
        ya, yb = clone(x, ei, ew, H2), cell(x, ei, ew, H2)
        assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-6) if weighted else torch.equal(ya, yb)


@pytest.mark.parametrize('improved', [False, True])
def test_tgcn_backward_matches_autograd_through_the_oracle(improved):
    """The reference trains this cell (examples/nodeproppred/tgcn.py:92: loss.backward() through tgcn.py:151-156).  Hand-written backward
    (nn/_tgcn_train.py, csrc/tgcn.hip) against torch autograd through the oracle restatement, two chained snapshots (the second one's
    loss reaches the first through the recurrent state H): every parameter gradient, d node_x and d H0 within 1e-4 of the gradient's max;
    and the grad-enabled forward agrees with the no-grad one."""
    from oracle.tgcn_ref import tgcn_cell_ref
    from tgm_amd.nn import TGCN

    torch.manual_seed(3 + improved)
    N, Fin, C = 255, 64, 128
    cell = TGCN(Fin, C, improved=improved).to(DEV).train()
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(0.05 * torch.randn_like(p))
    ref_p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in cell.state_dict().items()}
    snaps = []
    for s in range(2):
        E = 2500 + 700 * s
        ei = torch.randint(0, N, (2, E))
        ei[:, :5] = ei[0, :5]  # a few explicit self loops (they keep their weight)
        snaps.append((torch.randn(N, Fin), ei, torch.rand(E) + 0.1))
    H0 = 0.3 * torch.randn(N, C)
    wout = torch.randn(N, C)
    # oracle
    x_ref = [x.clone().requires_grad_(True) for x, _, _ in snaps]
    H0_ref = H0.clone().requires_grad_(True)
    H = H0_ref
    for (x, ei, ew), xr in zip(snaps, x_ref):
        H = tgcn_cell_ref(ref_p, xr, ei, ew, H, improved=improved)
    (H * wout).sum().backward()
    # device
    x_dev = [x.to(DEV).requires_grad_(True) for x, _, _ in snaps]
    H0_dev = H0.to(DEV).requires_grad_(True)
    Hd = H0_dev
    for (x, ei, ew), xd in zip(snaps, x_dev):
        Hd = cell(xd, ei.to(DEV), ew.to(DEV), Hd)
    close(Hd.detach().cpu(), H.detach(), 'forward (grad enabled)')
    with torch.no_grad():
        Hn = H0.to(DEV)
        for x, ei, ew in snaps:
            Hn = cell(x.to(DEV), ei.to(DEV), ew.to(DEV), Hn)
    # (same kernels in the same order; not asserted bit for bit: the dense adjacency accumulates duplicate edges with float atomics)
    close(Hn.cpu(), Hd.detach().cpu(), 'the saving forward vs the inference forward')
    (Hd * wout.to(DEV)).sum().backward()

    def gclose(got, ref, tag):
        scale = ref.abs().max().item()
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 1e-4 * max(scale, 1e-6), f'{tag}: max |d| {err:.3e} vs gradient max {scale:.3e}'

    for name, p in cell.named_parameters():
        assert p.grad is not None, name
        gclose(p.grad, ref_p[name].grad, name)
    for i, (xd, xr) in enumerate(zip(x_dev, x_ref)):
        gclose(xd.grad, xr.grad, f'd node_x[{i}]')
    gclose(H0_dev.grad, H0_ref.grad, 'd H0')


def test_gcn_conv_backward_matches_autograd_through_the_oracle():
    from oracle.tgcn_ref import gcn_conv_ref
    from tgm_amd.nn import GCNConv

    torch.manual_seed(11)
    N, Fin, C, E = 200, 24, 40, 1500
    conv = GCNConv(Fin, C).to(DEV)
    ei, ew, x = torch.randint(0, N, (2, E)), torch.rand(E) + 0.2, torch.randn(N, Fin)
    W, b, xr = (t.detach().cpu().clone().requires_grad_(True) for t in (conv.lin.weight, conv.bias, x))
    w = torch.randn(N, C)
    (gcn_conv_ref(xr, ei, ew, W, b) * w).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    out = conv(xd, ei.to(DEV), ew.to(DEV))
    (out * w.to(DEV)).sum().backward()
    for got, ref, tag in ((conv.lin.weight.grad, W.grad, 'lin.weight'), (conv.bias.grad, b.grad, 'bias'), (xd.grad, xr.grad, 'x')):
        err, scale = (got.cpu() - ref).abs().max().item(), ref.abs().max().item()
        assert err <= 1e-4 * scale, f'{tag}: {err:.3e} vs {scale:.3e}'


def test_discretize_on_device_matches_reference_golden_and_host_path():
    """DGData.discretize(device='cuda') -- every event group's grouping in tgmx_discretize_keep (csrc/discretize.hip) --
    against golden g9 (the reference's output; tgm/data/dg_data.py:423-564) and, bit for bit, against the host torch
    formulation on a trade-shaped stream (255 nodes, 468 k edges, seconds -> years) and on a stream whose int32 radix key
    wraps (ids ~ 1e6: src * base + dst overflows int32 exactly as in the reference)."""
    import os

    import numpy as np

    from tgm_amd import DGData

    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g9_discretize.npz'))
    T = torch.from_numpy
    d = DGData.from_raw(T(z['ts']), T(z['ei']), T(z['ex']), node_x_time=T(z['nt']), node_x_nids=T(z['nn']), node_x=T(z['nx']),
                        node_y_time=T(z['yt']), node_y_nids=T(z['yn']), node_y=T(z['yv']), time_delta='s')  # fmt: skip
    fields = ('time', 'edge_mask', 'edge_index', 'edge_x', 'node_x_mask', 'node_x_nids', 'node_x', 'node_y_mask', 'node_y_nids', 'node_y')

    def canon(times, *cols):  # events sharing a discretized timestamp: the reference's final argsort is not stable (dg_data.py:360)
        rows = np.concatenate([np.asarray(times, np.float64)[:, None]] + [np.asarray(c, np.float64).reshape(len(times), -1) for c in cols], 1)
        return rows[np.lexsort(rows.T[::-1])]

    for unit in ('m', 'h'):
        with pytest.warns(UserWarning):
            c = d.discretize(unit, device=DEV)
            h = d.discretize(unit)
        for f in fields:  # device path == host path, bit for bit
            assert torch.equal(getattr(c, f), getattr(h, f)), (unit, f)
        assert np.array_equal(c.time.numpy(), z[f'{unit}_time'])
        for mask, cols in (('edge_mask', ('edge_index', 'edge_x')), ('node_x_mask', ('node_x_nids', 'node_x')), ('node_y_mask', ('node_y_nids', 'node_y'))):
            got = canon(c.time[getattr(c, mask).long()].numpy(), *[getattr(c, f).numpy() for f in cols])
            exp = canon(z[f'{unit}_time'][z[f'{unit}_{mask}']], *[z[f'{unit}_{f}'] for f in cols])
            assert got.shape == exp.shape and np.array_equal(got, exp), f'{unit} {mask}'

    rng = np.random.default_rng(0)
    year = 365 * 24 * 3600
    for N, E, span in ((255, 468_000, 30 * year), (1_000_000, 300_000, 40 * year)):
        ts = torch.from_numpy(np.sort(rng.integers(0, span, E)))
        ei = torch.from_numpy(rng.integers(0, N, (E, 2)).astype(np.int32))
        if N > 100_000:
            ei[: E // 2] = ei[E // 2 : E // 2 * 2]  # repeated (src, dst) pairs so that groups exist despite the large id range
        data = DGData.from_raw(ts, ei, torch.rand(E, 1), time_delta='s')
        a, b = data.discretize('Y', device=DEV), data.discretize('Y')
        assert a.time.numel() < E
        for f in ('time', 'edge_mask', 'edge_index', 'edge_x'):
            assert torch.equal(getattr(a, f), getattr(b, f)), (N, f)
