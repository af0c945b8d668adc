import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import golden_util as gu
from test_tgcn_oracle_cpu import snapshots
from tgm_amd.nn import TGCN
from oracle.tgcn_ref import tgcn_cell_ref, gcn_conv_ref
meta, a = gu.load('g10_tgcn')
cell = TGCN(meta['Fin'], meta['C']).cuda().eval()
H = None; Ho = None
for s,(params, x, ei, ew, H_ref) in enumerate(snapshots(a, 'plain')):
    cell.load_state_dict(params)
    Hin = None if H is None else H.clone()
    H = cell(x.cuda(), ei.cuda(), None if ew is None else ew.cuda(), H)
    Ho = tgcn_cell_ref(params, x, ei, ew, Ho)
    print(s, 'gpu vs golden', (H.cpu()-H_ref).abs().max().item(), 'oracle vs golden', (Ho-H_ref).abs().max().item(), 'gpu vs oracle(gpu H in)', (H.cpu()-tgcn_cell_ref(params,x,ei,ew,None if Hin is None else Hin.cpu())).abs().max().item(), 'ew', ew is not None)
