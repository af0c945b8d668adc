"""TGN training path: gradients of the hand-written backward kernels (csrc/tgn_bwd.hip, nn/_tgn_train.py) against torch
autograd through the oracle (oracle/tgn_ref.py), with update_state called BEFORE backward like the reference's training
loop (examples/linkproppred/tgn.py:111-116).  TransformerConv is third-party: pinned to our restatement only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0) if torch.cuda.is_available() else None


def _setup(aggr, seed=0, dropout=False):
    from oracle.tgn_ref import TGNMemoryRef
    from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, MeanAggregator, TGNMemory

    rng = np.random.default_rng(seed)
    N, D, M, T_, bs = 40, 5, 12, 8, 24
    torch.manual_seed(seed)
    mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator() if aggr == 'last' else MeanAggregator()).to(DEV).train()
    enc = GraphAttentionEmbedding(M, 16, D, mem.time_enc).to(DEV).train()
    if not dropout:
        enc.conv.dropout = 0.0
    with torch.no_grad():  # non-trivial Time2Vec so that its gradient is exercised at these small time deltas
        mem.time_enc.w.weight.copy_(torch.rand(T_, 1) * 0.3)
        mem.time_enc.w.bias.copy_(torch.rand(T_))
    mp = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mem.state_dict().items() if k not in ('memory', 'last_update', '_assoc')}
    ep = {k: (mp['time_enc.w.' + k.split('.')[-1]] if k.startswith('time_enc.') else v.detach().cpu().clone().requires_grad_(True))
          for k, v in enc.state_dict().items()}  # fmt: skip
    ref = TGNMemoryRef(N, D, M, T_, mp, aggr)
    batches = []
    t0 = 10
    for b in range(3):
        src = torch.from_numpy(rng.integers(0, N, bs).astype(np.int32))
        dst = torch.from_numpy(rng.integers(0, N, bs).astype(np.int32))
        t = torch.from_numpy(np.sort(rng.integers(t0, t0 + 40, bs)).astype(np.int64))
        t0 += 40
        raw = torch.from_numpy(rng.random((bs, D), dtype=np.float32))
        batches.append((src, dst, t, raw))
    return mem, enc, ref, mp, ep, batches, rng, (N, D, M, T_)


@pytest.mark.parametrize('aggr,dropout', [('last', False), ('mean', False), ('last', True), ('mean', True)])
def test_tgn_parameter_gradients(aggr, dropout):
    """dropout=True: the reference's default GraphAttentionEmbedding (TransformerConv(dropout=0.1), tgn.py:25-27) in train mode;
    the oracle applies exactly the mask the device drew."""
    from oracle.tgn_ref import graph_attention_embedding_ref

    mem, enc, ref, mp, ep, batches, rng, (N, D, M, T_) = _setup(aggr, dropout=dropout)
    with torch.no_grad():
        for src, dst, t, raw in batches[:2]:
            mem.update_state(src.to(DEV), dst.to(DEV), t.to(DEV), raw.to(DEV))
            ref.update_state(src, dst, t, raw)
    src, dst, t, raw = batches[2]
    n_id = torch.unique(torch.cat([src, dst, torch.from_numpy(rng.integers(0, N, 10).astype(np.int32))])).to(torch.int32)
    U, E = n_id.numel(), 70
    edge_index = torch.from_numpy(rng.integers(0, U, (2, E)).astype(np.int64))
    e_t = torch.from_numpy(rng.integers(0, 130, E).astype(np.int64))
    e_x = torch.from_numpy(rng.random((E, D), dtype=np.float32))
    G = torch.from_numpy(rng.standard_normal((U, 16)).astype(np.float32))

    z, lu = mem(n_id.to(DEV))
    z2 = enc(z, lu, edge_index.to(DEV), e_t.to(DEV), e_x.to(DEV))
    loss = (z2 * G.to(DEV)).sum()
    mem.update_state(src.to(DEV), dst.to(DEV), t.to(DEV), raw.to(DEV))  # before backward, like the reference loop
    loss.backward()

    zr, lur = ref.forward(n_id.long())
    drop = (0.1, enc.conv._drop_seed, enc.conv._drop_calls) if dropout else None
    assert enc.conv.dropout == (0.1 if dropout else 0.0)
    z2r = graph_attention_embedding_ref(ep, zr, lur, edge_index, e_t, e_x, dropout=drop)
    if dropout:
        z2_off = graph_attention_embedding_ref(ep, zr, lur, edge_index, e_t, e_x).detach()
        assert not torch.allclose(z2_off, z2r.detach(), atol=1e-3), 'dropout had no effect'
    assert torch.equal(lu.cpu(), lur)
    assert ((z2.detach().cpu() - z2r.detach()).abs() <= 1e-5 * z2r.detach().abs().clamp(min=1)).all()
    (z2r * G).sum().backward()

    worst, report = ('', 0.0), []
    named = list(mem.named_parameters()) + [(f'enc.{k}', v) for k, v in enc.named_parameters() if not k.startswith('time_enc.')]
    refs = dict(mp)
    refs.update({f'enc.{k}': v for k, v in ep.items() if not k.startswith('time_enc.')})
    for name, p in named:
        r = refs[name].grad
        assert p.grad is not None and r is not None, name
        # lin_key.bias has a mathematically zero gradient (softmax is shift invariant per target): floor the scale
        err = ((p.grad.cpu() - r).abs().max() / r.abs().max().clamp(min=2e-3)).item()
        report.append((name, err, float(r.abs().max())))
        if err > worst[1]:
            worst = (name, err)
    assert worst[1] <= 2e-4, f'{aggr}: worst relative gradient error {worst}; all: {report}'


@pytest.mark.parametrize('dropout,fused', [(False, None), (True, None), (False, True)])
def test_tgn_training_step_reduces_loss(dropout, fused):
    """A few Adam steps through memory -> embedding on a fixed batch drive a regression loss down -- also with the
    reference-default constructor arguments (TransformerConv dropout 0.1), which used to raise."""
    mem, enc, ref, mp, ep, batches, rng, (N, D, M, T_) = _setup('last', seed=3, dropout=dropout)
    with torch.no_grad():
        for src, dst, t, raw in batches[:2]:
            mem.update_state(src.to(DEV), dst.to(DEV), t.to(DEV), raw.to(DEV))
    n_id = torch.arange(N, dtype=torch.int32, device=DEV)
    E = 120
    edge_index = torch.from_numpy(rng.integers(0, N, (2, E)).astype(np.int64)).to(DEV)
    e_t = torch.from_numpy(rng.integers(0, 100, E).astype(np.int64)).to(DEV)
    e_x = torch.from_numpy(rng.random((E, D), dtype=np.float32)).to(DEV)
    target = torch.randn(N, 16, device=DEV)
    params = list({id(p): p for p in list(mem.parameters()) + list(enc.parameters())}.values())
    opt = torch.optim.Adam(params, lr=1e-2, fused=fused)  # fused: the step does not bump Tensor._version (tgm_amd.nn._paramver)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        z, lu = mem(n_id)
        loss = ((enc(z, lu, edge_index, e_t, e_x) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert sum(losses[-5:]) < 0.8 * sum(losses[:5]), losses
