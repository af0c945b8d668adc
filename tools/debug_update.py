import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import _native
from oracle.ring_port import RingSamplerCPU
dev = torch.device('cuda', 0)
lib = _native.load()
N, E, D, tmax, B, bs = 4000, 15000, 3, 2_000_000, 6, 2500
rng = np.random.default_rng(1234 + N + E)
src = torch.from_numpy(rng.integers(0, N, E).astype(np.int32)); dst = torch.from_numpy(rng.integers(0, N, E).astype(np.int32))
ts = torch.from_numpy(np.sort(rng.integers(1, tmax, E)).astype(np.int64))
x = torch.from_numpy(rng.random((E, D), dtype=np.float32))
orc = RingSamplerCPU(N, [B], D, False, key_arith='int32')
ring = torch.zeros(N * B, 2, dtype=torch.int64, device=dev); wpos = torch.zeros(N, dtype=torch.int32, device=dev)
ring_x = torch.zeros(N * B, D, device=dev)
lib.tgmx_ring_reset(ring.data_ptr(), wpos.data_ptr(), B, N, _native.stream_ptr())
status = torch.zeros(1, dtype=torch.int32, device=dev)
for b in range(E // bs):
    lo, hi = b * bs, (b + 1) * bs
    n = hi - lo; m = 2 * n
    scratch = torch.empty(12 * m + 16, dtype=torch.int32, device=dev)
    s, d, t, xx = src[lo:hi].to(dev), dst[lo:hi].to(dev), ts[lo:hi].to(dev), x[lo:hi].to(dev)
    rc = lib.tgmx_ring_update(ring.data_ptr(), wpos.data_ptr(), ring_x.data_ptr(), D, B, N, s.data_ptr(), d.data_ptr(), t.data_ptr(),
                              xx.data_ptr(), n, lo, 0, 1, scratch.data_ptr(), status.data_ptr(), _native.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    orc.update(src[lo:hi], dst[lo:hi], ts[lo:hi], x[lo:hi])
    r = ring.cpu().numpy().view(np.int32).reshape(N, B, 4)
    ids, tt = r[:, :, 0], ring.cpu()[:, 1].reshape(N, B).numpy()
    bad_ids = np.argwhere(ids != orc.ids.numpy())
    bad_t = np.argwhere(tt != orc.times.numpy())
    wp = wpos.cpu().numpy() % B; owp = orc.wpos.numpy() % B
    bad_w = np.argwhere(wp != owp)
    print(f'batch {b}: bad ids {len(bad_ids)} bad times {len(bad_t)} bad wpos {len(bad_w)} status {int(status.item())}')
    if len(bad_ids) or len(bad_w):
        for node in list(dict.fromkeys([int(q[0]) for q in bad_ids] + [int(q[0]) for q in bad_w]))[:4]:
            sel = np.concatenate([np.where(src[lo:hi].numpy() == node)[0], n + np.where(dst[lo:hi].numpy() == node)[0]])
            print(' node', node, 'entries', sel.tolist(), 'got', ids[node].tolist(), tt[node].tolist(), 'wp', wp[node], 'ref', orc.ids[node].tolist(), orc.times[node].tolist(), 'wp', owp[node])
        break
