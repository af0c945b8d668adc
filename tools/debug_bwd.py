import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import golden_util as gu
from test_tgat_backward_gpu import _ref_grads
from tgm_amd.nn import TGAT
DEV='cuda'
for case in sys.argv[1:] or ['g5_tgat_one_layer', 'g5_tgat_small_nd8', 'g5_tgat_small_unix']:
    meta, params, inputs, z_ref = gu.tgat_case(case)
    enc = TGAT(edge_dim=meta['edge_dim'], num_layers=len(meta['num_nbrs']), dropout=0.0, **meta['dims']).to(DEV).train()
    enc.load_state_dict(params)
    dev = lambda v: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)
    z = enc(**{k: dev(v) for k, v in inputs.items()})
    torch.manual_seed(0); dz = torch.randn(z.shape)
    z.backward(dz.to(DEV))
    z64, g_ref = _ref_grads(params, meta['dims']['n_heads'], inputs, dz)
    print(case, 'fwd err', (z.detach().cpu()-z64).abs().max().item())
    for name, p in enc.named_parameters():
        g, r = p.grad.cpu(), g_ref[name]
        print(f'   {name:36s} rel err {((g-r).abs().max()/r.abs().max().clamp(min=1e-9)).item():.2e}  max|ref| {r.abs().max().item():.3e}')
