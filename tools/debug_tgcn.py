import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd.nn import GCNConv, tgcn, _ops
from oracle.tgcn_ref import gcn_conv_ref
torch.manual_seed(0)
N, Fin, C, E = 37, 5, 8, 90
ei = torch.randint(0, N, (2, E)); ei[:, :4] = ei[0, :4]
x = torch.randn(N, Fin)
A = tgcn.normalized_adjacency(ei.cuda(), None, N, 1.0, True).cpu()
# reference dense
src, dst = ei
loop = src == dst
w = torch.ones(E)
loop_w = torch.ones(N); 
s2 = torch.cat([src[~loop], torch.arange(N)]); d2 = torch.cat([dst[~loop], torch.arange(N)]); w2 = torch.cat([w[~loop], loop_w])
deg = torch.zeros(N).index_add_(0, d2, w2); dinv = deg.pow(-0.5)
Aref = torch.zeros(N, N).index_put_((d2, s2), dinv[s2]*w2*dinv[d2], accumulate=True)
print('A err', (A-Aref).abs().max().item())
conv = GCNConv(Fin, C).cuda()
out = conv(x.cuda(), ei.cuda()).cpu()
ref = gcn_conv_ref(x, ei, None, conv.lin.weight.detach().cpu(), conv.bias.detach().cpu())
print('conv err', (out-ref).abs().max().item())
xw = x @ conv.lin.weight.detach().cpu().T
xwt = torch.empty(C, N, device='cuda'); _ops.sgemm_nt(conv.lin.weight.detach(), x.cuda(), xwt)
print('xwt err', (xwt.cpu()-xw.T).abs().max().item())
Ag = tgcn.normalized_adjacency(ei.cuda(), None, N, 1.0, True)
o2 = torch.empty(N, C, device='cuda'); _ops.sgemm_nt(Ag, xwt, o2, K=N)
print('A@xw err', (o2.cpu() - Aref @ xw).abs().max().item(), 'strides', Ag.stride(), Ag.shape, Ag.data_ptr()%16)
