"""The oracle (oracle/recency_model.py) against the reference's own outputs.

This is what pins the oracle: every sampler golden under tests/golden/ was
produced by running the imported reference (tests/golden/make_golden.py).
Both formulations -- streaming history and static CSR index -- must reproduce
ids, timestamps and feature rows bit-for-bit.
"""
import hashlib

import numpy as np
import pytest

import golden_util as gu
import torch

from oracle.recency_model import CsrModel, HistoryModel
from oracle.ring_port import RingSamplerCPU

# fixtures recorded in the regime where the reference's int32 composite key wraps
# (node * (max_time+1) >= 2**31, recency.py:347): only the faithful ring port matches there
WRAPPING = {'g3_wiki_medium'}


def _check_hops(meta, a, b, hops, tag):
    for h, (sn, stt, nn, nt, nx) in enumerate(hops):
        for key, val in (('seed_nids', sn), ('seed_times', stt), ('nbr_nids', nn), ('nbr_edge_time', nt), ('nbr_edge_x', nx)):
            name = f'b{b}_h{h}_{key}'
            if meta.get('digest_only'):
                assert hashlib.sha256(np.ascontiguousarray(val).tobytes()).hexdigest() == meta['digests'][name], f'{tag} {name}'
                if name in a:
                    np.testing.assert_array_equal(val, a[name], err_msg=f'{tag} {name}')
            else:
                exp = a[name]
                assert val.dtype == exp.dtype, f'{tag} {name} dtype {val.dtype} vs {exp.dtype}'
                assert val.shape == exp.shape, f'{tag} {name} shape {val.shape} vs {exp.shape}'
                np.testing.assert_array_equal(val, exp, err_msg=f'{tag} {name}')


@pytest.mark.parametrize('case', gu.sampler_cases())
def test_history_model_matches_reference(case):
    meta, a = gu.load(case)
    edge_x = a.get('edge_x')
    model = HistoryModel(meta['num_nodes'], meta['num_nbrs'], meta['directed'])
    b = 0
    for kind, lo, hi in gu.schedule(meta):
        if kind == 'reset':
            model.reset()
            continue
        seeds, times = gu.seeds_for(meta, a, lo, hi)
        hops = model.step(seeds, times, a['src'][lo:hi], a['dst'][lo:hi], a['ts'][lo:hi], lo, edge_x)
        _check_hops(meta, a, b, hops, case)
        b += 1
    assert b == meta['num_batches']


@pytest.mark.parametrize('case', gu.sampler_cases())
def test_csr_model_matches_reference(case):
    meta, a = gu.load(case)
    edge_x = a.get('edge_x')
    model = CsrModel(a['src'], a['dst'], a['ts'], meta['num_nodes'], gu.batch_starts(meta), meta['directed'])
    b = 0
    for kind, lo, hi in gu.schedule(meta):
        if kind == 'reset':
            continue  # every epoch in the fixtures restarts at edge 0, so ev_lo stays 0
        seeds, times = gu.seeds_for(meta, a, lo, hi)
        hops = model.step(seeds, times, meta['num_nbrs'], 0, lo, edge_x)
        _check_hops(meta, a, b, hops, case)
        b += 1
    assert b == meta['num_batches']


@pytest.mark.parametrize('case', gu.sampler_cases() + ['g3_wiki_medium'])
def test_ring_port_matches_reference(case):
    """The faithful tensor-program port: bit-exact on EVERY fixture, wrapping regime included."""
    meta, a = gu.load(case)
    if case == 'g3_wiki_medium':
        from tgm_amd.synth import make_stream

        edge_x = make_stream('wiki', seed=1337, num_edges=20_000, edge_dim=8).edge_x
    else:
        edge_x = torch.from_numpy(a['edge_x']) if 'edge_x' in a else None
    D = 0 if edge_x is None else edge_x.shape[1]
    model = RingSamplerCPU(meta['num_nodes'], meta['num_nbrs'], D, meta['directed'])
    T = torch.from_numpy
    b = 0
    for kind, lo, hi in gu.schedule(meta):
        if kind == 'reset':
            model.reset()
            continue
        seeds, times = gu.seeds_for(meta, a, lo, hi)
        hops = model.step(T(seeds), T(times), T(a['src'][lo:hi]), T(a['dst'][lo:hi]), T(a['ts'][lo:hi]),
                          None if edge_x is None else edge_x[lo:hi])
        _check_hops(meta, a, b, [tuple(t.numpy() for t in h) for h in hops], case)
        b += 1
    assert b == meta['num_batches']


def test_intended_order_differs_only_when_key_wraps():
    """On g3 the reference (int32 key) and the intended (node, time) order disagree;
    with key_arith='int64' the port equals the history / CSR models instead."""
    from tgm_amd.synth import make_stream

    meta, a = gu.load('g3_wiki_medium')
    st = make_stream('wiki', seed=1337, num_edges=20_000, edge_dim=8)
    T = torch.from_numpy
    port = RingSamplerCPU(meta['num_nodes'], meta['num_nbrs'], 8, meta['directed'], key_arith='int64')
    csr = CsrModel(a['src'], a['dst'], a['ts'], meta['num_nodes'], gu.batch_starts(meta), meta['directed'])
    ex = st.edge_x.numpy()
    for b, (kind, lo, hi) in enumerate(gu.schedule(meta)):
        seeds, times = gu.seeds_for(meta, a, lo, hi)
        hops = port.step(T(seeds), T(times), T(a['src'][lo:hi]), T(a['dst'][lo:hi]), T(a['ts'][lo:hi]), st.edge_x[lo:hi])
        if b in (3, 40, 99):
            ref = csr.step(seeds, times, meta['num_nbrs'], 0, lo, ex)
            for h in range(2):
                for i in (2, 3, 4):
                    np.testing.assert_array_equal(hops[h][i].numpy(), ref[h][i])


def test_csr_model_wiki_medium_prefix():
    """20k-edge wiki-shaped stream: the intended-order CSR model equals the reference
    until the first batch whose update key wraps into a collision (checked: early batches)."""
    from tgm_amd.synth import make_stream

    meta, a = gu.load('g3_wiki_medium')
    st = make_stream('wiki', seed=1337, num_edges=20_000, edge_dim=8)
    np.testing.assert_array_equal(st.src.numpy(), a['src'])
    np.testing.assert_array_equal(st.ts.numpy(), a['ts'])
    edge_x = st.edge_x.numpy()
    model = CsrModel(a['src'], a['dst'], a['ts'], meta['num_nodes'], gu.batch_starts(meta), meta['directed'])
    for b, (kind, lo, hi) in enumerate(gu.schedule(meta)):
        if b not in (0, 1, 2):
            continue
        seeds, times = gu.seeds_for(meta, a, lo, hi)
        hops = model.step(seeds, times, meta['num_nbrs'], 0, lo, edge_x)
        _check_hops(meta, a, b, hops, 'g3')
