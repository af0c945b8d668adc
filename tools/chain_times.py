#!/usr/bin/env python
"""Diagnostic: per-phase cycle stamps of tgat_chain64_kernel (temporary)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tgm_amd import _native
from tgm_amd.nn import TGAT
from tgm_amd.synth import make_stream

stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
BS = int(os.environ.get('BS', '200'))
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, BS, [20, 20], 'ring', dev)
enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(dev).eval()
starts = loader._starts
lib = _native.load()
lib.tgmx_debug_chain_times.argtypes = [ctypes.c_void_p]
lib.tgmx_debug_chain_times.restype = None
with hm.activate('bench'), torch.no_grad():
    for i in range(300 * 200 // BS):
        b = loader(starts[i])
    for _ in range(5):
        z = enc(dg.static_node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    torch.cuda.synchronize()
    buf = torch.zeros(400 * 4 * 8, dtype=torch.int64, device=dev)
    lib.tgmx_debug_chain_times(buf.data_ptr())
    z = enc(dg.static_node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    torch.cuda.synchronize()
    lib.tgmx_debug_chain_times(None)
t = buf.view(-1, 8).cpu().double()
t = t[t[:, 0] > 0]
base = t[:, 0].min()
names = ['init', 'stage1 (2 heads)', 'stage2 W_O gemm', 'LN epilogue', 'fc1', 'fc2']
print('waves', t.shape[0], 'kernel span (cycles of the counter):', float(t[:, 6].max() - base))
print('start skew: mean %.0f max %.0f' % (float((t[:, 0] - base).mean()), float((t[:, 0] - base).max())))
for i, n in enumerate(names):
    dlt = t[:, i + 1] - t[:, i]
    print('%-18s mean %8.0f  min %8.0f  max %8.0f' % (n, float(dlt.mean()), float(dlt.min()), float(dlt.max())))
print('total per wave mean', float((t[:, 6] - t[:, 0]).mean()))
print('fc2 gemm %.0f  fc2 store epilogue %.0f' % (float((t[:, 7] - t[:, 5]).mean()), float((t[:, 6] - t[:, 7]).mean())))

if int(os.environ.get('TGMX_C64_MODE', '0')) & 8:
    print('inside the chunk loops, summed over all GEMMs (cycles per wave): issue+xg %.0f  ds_read+mfma %.0f  xfetch %.0f  wait+barrier %.0f' % (
        float(t[:, 7].mean()), float(t[:, 3].mean()), float(t[:, 1].mean()), float(t[:, 2].mean())))
