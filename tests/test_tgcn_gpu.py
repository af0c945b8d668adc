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
