"""The TGN-memory oracle against the reference's TGNMemory outputs (goldens g8): bit-level
store semantics, 1e-5 relative on the memory values (fp32 GRU / Time2Vec)."""
import pytest
import torch

import golden_util as gu
from oracle.tgn_ref import TGNMemoryRef

RTOL = 1e-5
CASES = ['g8_tgn_last', 'g8_tgn_mean', 'g8_tgn_last_unix']


def close(got, ref, tag):
    err = (got - ref).abs()
    worst = (err / (RTOL * ref.abs().clamp(min=1.0))).max().item() if ref.numel() else 0.0
    assert worst <= 1.0, f'{tag}: {worst:.2f}x the 1e-5 bound'


def drive(meta, a, mem, to=lambda t: t):
    """Replay the fixture's schedule on `mem` (oracle or product), yielding what to compare."""
    T = torch.from_numpy
    bs, nb, ntrain = meta['batch_size'], meta['num_batches'], meta['train_batches']
    E = len(a['src'])
    for b in range(nb):
        lo, hi = b * bs, min((b + 1) * bs, E)
        if b == ntrain:
            mem.eval()
            yield 'flush', b
        z, lu = mem.forward(to(T(a[f'b{b}_n_id'])))
        yield 'fwd', b, z, lu
        mem.update_state(to(T(a['src'][lo:hi])), to(T(a['dst'][lo:hi])), to(T(a['ts'][lo:hi])), to(T(a['raw'][lo:hi])))
        if f'b{b}_memory' in a:
            yield 'state', b


@pytest.mark.parametrize('case', CASES)
def test_tgn_memory_oracle_matches_reference(case):
    meta, a = gu.load(case)
    T = torch.from_numpy
    params = {k[2:]: T(v) for k, v in a.items() if k.startswith('w_')}
    mem = TGNMemoryRef(meta['num_nodes'], meta['raw_msg_dim'], meta['memory_dim'], meta['time_dim'], params, meta['aggr'])
    for ev in drive(meta, a, mem):
        if ev[0] == 'fwd':
            _, b, z, lu = ev
            close(z, T(a[f'b{b}_z']), f'{case} b{b} z')
            assert torch.equal(lu, T(a[f'b{b}_last_update'])), f'{case} b{b} last_update'
        elif ev[0] == 'state':
            b = ev[1]
            close(mem.memory, T(a[f'b{b}_memory']), f'{case} b{b} memory')
            assert torch.equal(mem.last_update, T(a[f'b{b}_mem_last_update']))
        else:
            close(mem.memory, T(a['flush_memory']), f'{case} flush')
            assert torch.equal(mem.last_update, T(a['flush_last_update']))
