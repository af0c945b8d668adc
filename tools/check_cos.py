#!/usr/bin/env python
"""Accuracy of the device cosine (cos_t2v, csrc/common.h) against float64 cos, range by range -- every range is a
separate launch so that whole waves take the float (|x| < 8e6) or the double reduction path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import _native
if os.environ.get("TGMX_LIB"):
    _native.load(os.environ["TGMX_LIB"])
from tgm_amd.nn import _ops
torch.manual_seed(0)
w = torch.ones(1, device='cuda'); b = torch.zeros(1, device='cuda')
worst = 0.0
for name, x in [('[0, 10)', torch.rand(400000) * 10), ('[0, 3e6)', torch.rand(400000) * 3e6), ('[3e6, 8e6) float path edge', 3e6 + torch.rand(400000) * 4.999e6),
                ('[-8e6, 0)', -torch.rand(400000) * 7.999e6), ('[8e6, 2.1e9) double path', 8e6 + torch.rand(400000) * 2.1e9),
                ('specials', torch.tensor([0., 1e-30, 3.14159265, 1.5707963, 7999999.5, -7999999.5, 8000000.0, 2147483648.]))]:
    x = x.float().cuda()
    got = _ops.time2vec(x, w, b)[:, 0].cpu().double()
    err = (got - torch.cos(x.cpu().double())).abs()
    worst = max(worst, err.max().item())
    print(f'{name:32s} max abs err {err.max().item():.3e} at x={x.cpu()[err.argmax()].item():.1f}  mean {err.mean().item():.2e}')
print('max abs err vs f64 cos:', worst)
assert worst < 2.5e-7
