"""TGAT training path: parameter gradients of the HIP backward vs torch autograd on the CPU restatement
(the folded algorithm).
Bar: |g - g_ref| <= 1e-4 * max|g_ref| per parameter tensor (fp32 kernels, sums over up to ~250k terms)."""
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ref_grads(params, n_heads, inputs, dz, dtype, dropout=None, want_z=False):
    """torch autograd (CPU) through the folded restatement in `dtype`.  Time2Vec is evaluated at the float32
    argument the model uses (float32(dt), one fma rounded to float32; straight-through for the roundings):
    at unix-scale timestamps cos / sin are only meaningful there."""
    import oracle.tgat_ref as tr
    from oracle import tgat_fold

    def t2v(t, w, b):
        arg = t.unsqueeze(-1).float().double() * w.reshape(-1).double() + b.double()
        return torch.cos(arg + (arg.float().double() - arg).detach()).to(dtype)

    p = {k: v.to(dtype).requires_grad_(True) for k, v in params.items()}
    f = lambda t: t.to(dtype) if t.is_floating_point() else t
    old = tr.time2vec
    tr.time2vec = tgat_fold.time2vec = t2v
    try:
        z = tgat_fold.tgat_forward_folded(p, n_heads, f(inputs['node_x']), inputs['seed_nids'], inputs['seed_times'], inputs['nbr_nids'],
                                          [f(x) for x in inputs['nbr_edge_x']], inputs['nbr_edge_time'], dropout=dropout)  # fmt: skip
    finally:
        tr.time2vec = tgat_fold.time2vec = old
    (z * dz.to(dtype)).sum().backward()
    grads = {k: v.grad.float() for k, v in p.items()}
    return (grads, z.detach().float()) if want_z else grads


def _worst(enc, g_ref):
    worst = ('', 0.0)
    for name, p in enc.named_parameters():
        r = g_ref[name]
        err = ((p.grad.cpu() - r).abs().max() / r.abs().max().clamp(min=1e-6)).item()
        if err > worst[1]:
            worst = (name, err)
    return worst


@pytest.mark.parametrize('case', ['g5_tgat_small_unix', 'g5_tgat_small_nd8', 'g5_tgat_one_layer', 'g5_tgat_example_dims'])
def test_tgat_parameter_gradients(case):
    from tgm_amd.nn import TGAT

    meta, params, inputs, z_ref = gu.tgat_case(case)
    enc = TGAT(edge_dim=meta['edge_dim'], num_layers=len(meta['num_nbrs']), dropout=0.0, **meta['dims']).to(DEV).train()
    enc.load_state_dict(params)
    dev = lambda v: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)
    z = enc(**{k: dev(v) for k, v in inputs.items()})
    assert z.requires_grad
    torch.manual_seed(0)
    dz = torch.randn(z.shape)
    z.backward(dz.to(DEV))
    assert ((z.detach().cpu() - z_ref).abs() <= 1e-5 * z_ref.abs().clamp(min=1)).all()
    assert all(p.grad is not None for p in enc.parameters())
    # The merge layer's ReLU makes the gradient discontinuous: a pre-activation that is ~1e-7 gets a different
    # mask in fp32 and fp64 (torch's own fp32-vs-fp64 autograd differ by 4e-3 on the example-dims case from
    # exactly that), so the kernels must agree with the fp64 OR the fp32 autograd reference on every parameter.
    w64 = _worst(enc, _ref_grads(params, meta['dims']['n_heads'], inputs, dz, torch.float64))
    w32 = _worst(enc, _ref_grads(params, meta['dims']['n_heads'], inputs, dz, torch.float32)) if w64[1] > 1e-4 else w64
    assert min(w64[1], w32[1]) <= 1e-4, f'{case}: worst rel err vs fp64 {w64}, vs fp32 {w32}'


@pytest.mark.parametrize('case,dropout', [('g5_tgat_small_nd8', 0.0), ('g5_tgat_small_nd8', 0.1), ('g5_tgat_example_dims', 0.1)])
def test_one_call_backward_equals_the_composed_one(case, dropout, monkeypatch):
    """``tgmx_tgat_backward`` (the whole backward as one native call) runs the kernels the Python composition launches one by one, in
    its order: every parameter gradient is bit-identical, the Time2Vec bias to one rounding of a sine (``TGMX_TGAT_BWD=py`` selects the
    composition)."""
    from tgm_amd.nn import TGAT

    meta, params, inputs, _ = gu.tgat_case(case)
    dev = lambda v: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)
    args = {k: dev(v) for k, v in inputs.items()}

    def grads(mode):
        monkeypatch.setenv('TGMX_TGAT_BWD', mode)
        torch.manual_seed(7)
        enc = TGAT(edge_dim=meta['edge_dim'], num_layers=2, dropout=dropout, **meta['dims']).to(DEV).train()
        enc.load_state_dict(params)
        z = enc(**args)
        z.backward(torch.cos(torch.arange(z.numel(), device=DEV, dtype=torch.float32)).view_as(z))
        return z.detach().clone(), {n: p.grad.clone() for n, p in enc.named_parameters()}

    (za, a), (zb, b) = grads('py'), grads('native')
    assert torch.equal(za, zb) and a.keys() == b.keys() and len(a) == 22
    diff = {n: float((a[n] - b[n]).abs().max()) for n in a if a[n].shape != b[n].shape or not torch.equal(a[n], b[n])}
    # the one arithmetic difference: d tb -= sin(tb) * g evaluates sin in our kernel instead of torch's (<= 1 ulp of the term)
    tb = diff.pop('time_encoder.w.bias', 0.0)
    assert not diff and tb <= 4e-7 * float(a['time_encoder.w.bias'].abs().max()), (diff, tb)
    assert all(float(v.abs().max()) > 0 for v in a.values())


def test_one_call_backward_at_the_headline_shape(monkeypatch):
    """600 seeds, k = [20, 20] (12 600 attention rows in layer 1: the weight-gradient GEMMs' split partials are megabytes -- a
    scratch area sized for the golden cases only would be overrun here): native == composed, sampler outputs from the HIP sampler."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.nn import TGAT
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=11, num_edges=30_000)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x, static_node_x=st.node_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(8227, st.num_nodes, seed=3))
    hm.register('k', RecencyNeighborHook(st.num_nodes, [20, 20], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']))
    with hm.activate('k'):
        for b, batch in enumerate(DGDataLoader(dg, batch_size=200, hook_manager=hm, output_pool=0)):
            if b == 120:
                break
    args = (dg.static_node_x, batch.seed_nids, batch.seed_times, batch.nbr_nids, batch.nbr_edge_x, batch.nbr_edge_time)

    def grads(mode):
        monkeypatch.setenv('TGMX_TGAT_BWD', mode)
        torch.manual_seed(7)
        enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(DEV).train()  # the reference-default dropout 0.1
        z = enc(*args)
        ((z[:200] * z[200:400]).sum(-1).sigmoid().mean() - (z[:200] * z[400:]).sum(-1).sigmoid().mean()).backward()
        return {n: p.grad.clone() for n, p in enc.named_parameters()}

    a, b = grads('py'), grads('native')
    diff = {n: float((a[n] - b[n]).abs().max()) for n in a if not torch.equal(a[n], b[n])}
    tb = diff.pop('time_encoder.w.bias', 0.0)
    assert not diff and tb <= 4e-7 * float(a['time_encoder.w.bias'].abs().max()), (diff, tb)
    assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in b.values())


def test_training_with_edge_features_by_id_equals_the_dense_copies():
    """``RecencyNeighborHook(edge_features='by_id')`` in TRAINING: the saving forward's attention and the attention backward inside
    ``tgmx_tgat_backward`` read the rows of the resident store by edge id instead of the sampler's dense [rows, k, D] copies -- same
    values in the same order: embeddings and every parameter gradient bit-identical (dropout 0.1, the headline shape)."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.core import EdgeFeaturesById
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.nn import TGAT
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=11, num_edges=30_000)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x, static_node_x=st.node_x), device=DEV)

    def run(features):
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(8227, st.num_nodes, seed=3))
        hm.register('k', RecencyNeighborHook(st.num_nodes, [20, 20], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'],
                                             edge_features=features))  # fmt: skip
        with hm.activate('k'):
            for b, batch in enumerate(DGDataLoader(dg, batch_size=200, hook_manager=hm, output_pool=0)):
                if b == 120:
                    break
        assert isinstance(batch.nbr_edge_x, EdgeFeaturesById) == (features == 'by_id')
        torch.manual_seed(7)
        enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(DEV).train()
        z = enc(dg.static_node_x, batch.seed_nids, batch.seed_times, batch.nbr_nids, batch.nbr_edge_x, batch.nbr_edge_time)
        ((z[:200] * z[200:400]).sum(-1).sigmoid().mean() - (z[:200] * z[400:]).sum(-1).sigmoid().mean()).backward()
        assert not getattr(enc, '_by_id_unsupported', False)
        return z.detach().clone(), {n: p.grad.clone() for n, p in enc.named_parameters()}

    (za, a), (zb, b) = run('dense'), run('by_id')
    assert torch.equal(za, zb)
    diff = {n: float((a[n] - b[n]).abs().max()) for n in a if not torch.equal(a[n], b[n])}
    assert not diff, diff


def _train_losses(make_opt, steps=40):
    from tgm_amd.nn import TGAT, invalidate_parameter_caches

    meta, params, inputs, _ = gu.tgat_case('g5_tgat_small_nd8')
    enc = TGAT(edge_dim=meta['edge_dim'], num_layers=2, dropout=0.0, **meta['dims']).to(DEV).train()
    enc.load_state_dict(params)
    dev = lambda v: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)
    args = {k: dev(v) for k, v in inputs.items()}
    torch.manual_seed(1)
    target = torch.randn(30, meta['dims']['embed_dim'], device=DEV)
    opt = make_opt(enc.parameters())
    losses = []
    for _ in range(steps):
        enc.zero_grad()
        loss = ((enc(**args) - target) ** 2).mean()
        loss.backward()
        if opt is not None:
            opt.step()
        else:  # hand-rolled SGD through .data: invisible to autograd's version counter
            for p in enc.parameters():
                if p.grad is not None:
                    p.data.add_(p.grad, alpha=-1e-2)
            invalidate_parameter_caches()
        losses.append(float(loss.detach()))
    return losses


def test_tgat_training_step_reduces_loss():
    """A few Adam steps on a fixed batch with the HIP forward/backward must drive a regression loss down."""
    losses = _train_losses(lambda ps: torch.optim.Adam(ps, lr=1e-2))
    assert losses[-1] < 0.8 * losses[0] and all(b < a * 1.05 for a, b in zip(losses, losses[1:])), losses


@pytest.mark.parametrize('variant', ['adam_fused', 'sgd_fused', 'data_writes'])
def test_training_sees_updates_that_do_not_bump_tensor_versions(variant):
    """The fused optimizers update the parameters without bumping ``Tensor._version``; so do writes through ``p.data``.  The kernels'
    padded weight copies are cached against tgm_amd.nn._paramver.param_key (counts optimizer steps; ``invalidate_parameter_caches()``
    for hand-rolled updates): the loss trajectory must be the plain optimizer's, not that of a model that never sees its updates."""
    plain, other = {
        'adam_fused': (lambda ps: torch.optim.Adam(ps, lr=1e-2), lambda ps: torch.optim.Adam(ps, lr=1e-2, fused=True)),
        'sgd_fused': (lambda ps: torch.optim.SGD(ps, lr=1e-2), lambda ps: torch.optim.SGD(ps, lr=1e-2, fused=True)),
        'data_writes': (lambda ps: torch.optim.SGD(ps, lr=1e-2), lambda ps: None),
    }[variant]
    a, b = _train_losses(plain, 25), _train_losses(other, 25)
    assert a[-1] < 0.97 * a[0], a  # the plain run does learn ...
    # ... and the other one follows it (the fused kernels round differently and training amplifies it; a model
    # that never saw its updates would sit at a[0])
    tol = 2e-3 if variant == 'data_writes' else 5e-2
    assert all(abs(x - y) <= tol * abs(x) for x, y in zip(a, b)) and b[-1] < 0.97 * b[0], (variant, a, b)


@pytest.mark.parametrize('case', ['g5_tgat_small_nd8', 'g5_tgat_example_dims'])
def test_tgat_train_mode_dropout_forward_and_gradients(case):
    """The reference's DEFAULT training configuration (dropout = 0.1 on the attention weights and on the W_O output,
    attention.py:119,126; tgat.py:67): forward and every parameter gradient against torch autograd through the oracle
    run with exactly the masks the device drew (oracle/dropout_ref.py restates the counter-based generator)."""
    from tgm_amd.nn import TGAT

    meta, params, inputs, z_eval = gu.tgat_case(case)
    enc = TGAT(edge_dim=meta['edge_dim'], num_layers=len(meta['num_nbrs']), **meta['dims']).to(DEV).train()  # dropout: the default 0.1
    assert enc.attn[0].dropout.p == pytest.approx(0.1)
    enc.load_state_dict(params)
    dev = lambda v: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)
    args = {k: dev(v) for k, v in inputs.items()}
    torch.manual_seed(0)
    dz = torch.randn(z_eval.shape)
    for call in (1, 2):  # a fresh mask per call
        enc.zero_grad()
        z = enc(**args)
        z.backward(dz.to(DEV))
        assert enc._drop_calls == call
        drop = (0.1, enc._drop_seed, call)
        g64, z64 = _ref_grads(params, meta['dims']['n_heads'], inputs, dz, torch.float64, dropout=drop, want_z=True)
        assert ((z.detach().cpu() - z64).abs() <= 1e-5 * z64.abs().clamp(min=1)).all(), 'train-mode forward'
        assert not torch.allclose(z.detach().cpu(), z_eval, atol=1e-3), 'dropout had no effect'
        w64 = _worst(enc, g64)
        w32 = _worst(enc, _ref_grads(params, meta['dims']['n_heads'], inputs, dz, torch.float32, dropout=drop)) if w64[1] > 1e-4 else w64
        assert min(w64[1], w32[1]) <= 1e-4, f'{case} call {call}: worst rel err vs fp64 {w64}, vs fp32 {w32}'
    # eval mode is untouched by all of this
    z_e = enc.eval()(**args)
    assert ((z_e.detach().cpu() - z_eval).abs() <= 1e-5 * z_eval.abs().clamp(min=1)).all()


def test_dropout_mask_statistics_and_determinism():
    """tgmx_dropout: kept fraction ~ 1 - p, kept values scaled by 1 / (1 - p), same (seed, stream) -> same mask, another
    stream -> another mask; bit-identical to the numpy restatement."""
    from oracle.dropout_ref import dropout_scale
    from tgm_amd import _native

    lib = _native.load()
    R, C, p = 4096, 77, 0.1
    x = torch.ones((R, C), device=DEV)
    outs = []
    for stream in (5, 5, 6):
        y = torch.empty_like(x)
        _native.check(lib.tgmx_dropout(x.data_ptr(), C, R, C, _native.dropout_desc(p, 1234, stream), y.data_ptr(), C, _native.stream_ptr()), 'tgmx_dropout')
        outs.append(y.cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0], dropout_scale(p, 1234, 5, (R, C)))
    kept = (outs[0] != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.005 and torch.allclose(outs[0][outs[0] != 0], torch.tensor(1 / 0.9))


def test_tgat_default_constructor_trains():
    """the reference-default constructor arguments (dropout 0.1) must train, not raise."""
    from tgm_amd.nn import TGAT

    meta, params, inputs, _ = gu.tgat_case('g5_tgat_small_nd8')
    enc = TGAT(edge_dim=meta['edge_dim'], num_layers=2, **meta['dims']).to(DEV).train()
    enc.load_state_dict(params)
    dev = lambda v: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)
    args = {k: dev(v) for k, v in inputs.items()}
    torch.manual_seed(1)
    target = torch.randn(30, meta['dims']['embed_dim'], device=DEV)
    opt = torch.optim.Adam(enc.parameters(), lr=1e-2)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = ((enc(**args) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert sum(losses[-10:]) < 0.8 * sum(losses[:10]), losses
    with torch.no_grad():  # train mode under no_grad still applies dropout, like nn.Dropout
        a, b = enc(**args), enc(**args)
    assert not torch.equal(a, b)
