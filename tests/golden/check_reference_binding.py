#!/usr/bin/env python
"""Reference-side binding check, run in the build container only.

Imports the REFERENCE (/root/reference; PyG replaced by tests/golden/_pyg_stub) and checks, against the reference's own
code, the claims a maintainer relies on when dropping our hooks into the reference's ``HookManager``:

  1. ``tgm_amd.hooks.RecencyNeighborHook`` / ``RandomNegativeEdgeSamplerHook`` / ``DeduplicationHook`` / ``NeighborSamplerHook``
     satisfy the reference's runtime-checkable ``DGHook`` protocol (tgm/hooks/base.py:10-24; checked by
     ``HookManager._ensure_valid_hook``, hook_manager.py:373-377) and register without error;
  2. the reference manager orders negatives -> our neighbor sampler through its implicit edge
     (hook_manager.py:427-430: ``'neg' in produces`` before ``'nbr_nids' in produces``), whatever the registration order,
     also when the negative hook is the reference's own;
  3. our hooks' ``requires`` / ``produces`` equal the reference hooks' for the same constructor arguments;
  4. our ``DGBatch`` has the reference's fields (names, order, defaults) and our ``HookManager`` resolves the same order;
  5. ``state_dict`` keys / shapes of our ``TGAT`` equal the reference's for the example configuration;
  6. the reference's own host-side unit tests (test_dataloader / test_hook_manager / test_registry / test_dgraph /
     test_deduplication_hook / test_seed, read in place) pass with ``tgm`` resolving to ``tgm_amd``
     (``replay_reference_tests.py``, its own process: this one has the real ``tgm`` imported).

Prints one line per check and exits non-zero on the first failure.  No kernel runs (no GPU here).

    python tests/golden/check_reference_binding.py
"""
from __future__ import annotations

import dataclasses
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, '_pyg_stub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, REPO)

import tgm  # noqa: E402
import tgm.hooks as ref_hooks  # noqa: E402
from tgm.hooks.base import DGHook as RefDGHook  # noqa: E402

import tgm_amd  # noqa: E402
import tgm_amd.hooks as our_hooks  # noqa: E402


def ok(msg: str) -> None:
    print(f'[binding] ok: {msg}', flush=True)


def main() -> None:
    N, ks = 50, [3, 2]
    keys, tkeys = ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']
    ours = {
        'nbr': our_hooks.RecencyNeighborHook(N, ks, keys, tkeys),
        'neg': our_hooks.RandomNegativeEdgeSamplerHook(0, N),
        'dedup': our_hooks.DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']),
        'uniform': our_hooks.NeighborSamplerHook(ks, keys, tkeys),
    }
    theirs = {
        'nbr': ref_hooks.RecencyNeighborHook(N, ks, keys, tkeys),
        'neg': ref_hooks.RandomNegativeEdgeSamplerHook(0, N),
        'dedup': ref_hooks.DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']),
        'uniform': ref_hooks.NeighborSamplerHook(ks, keys, tkeys),
    }
    # 1. protocol
    for name, h in ours.items():
        assert isinstance(h, RefDGHook), f'{name}: not a reference DGHook'
    ok('our hooks satisfy the reference DGHook protocol (tgm/hooks/base.py:10-24)')
    # 3. same requires / produces
    for name in ours:
        assert ours[name].requires == theirs[name].requires, (name, ours[name].requires, theirs[name].requires)
        assert ours[name].produces == theirs[name].produces, (name, ours[name].produces, theirs[name].produces)
        # class attribute, as the reference's own tests read it (test_recency_nbr_hook.py:111); the reference INSTANCE reports
        # False for RecencyNeighborHook (its SeedableHook dataclass base re-declares the field) -- nothing reads it
        assert type(ours[name]).has_state == type(theirs[name]).has_state, name
    ok('requires / produces / has_state equal the reference hooks\' for the same constructor arguments')
    # 2. the reference manager accepts them and keeps neg -> nbr, in both registration orders and with mixed providers
    for label, neg_hook in (('our negatives', ours['neg']), ('reference negatives', theirs['neg'])):
        for order in (('nbr', 'neg', 'dedup'), ('dedup', 'neg', 'nbr'), ('neg', 'nbr', 'dedup')):
            hm = ref_hooks.HookManager(keys=['train'])
            for name in order:
                hm.register('train', neg_hook if name == 'neg' else ours[name])
            hm.resolve_hooks('train')
            resolved = hm._key_to_hooks['train']
            pos = {id(h): i for i, h in enumerate(resolved)}
            assert pos[id(neg_hook)] < pos[id(ours['nbr'])] < pos[id(ours['dedup'])], (label, order, resolved)
    ok('reference HookManager: registers our hooks, resolves negatives -> neighbor sampler -> dedup (hook_manager.py:373-377, 427-430)')
    # 4. batch record + our manager resolves the same order
    ref_fields = [(f.name, f.default) for f in dataclasses.fields(tgm.DGBatch)]
    our_fields = [(f.name, f.default) for f in dataclasses.fields(tgm_amd.DGBatch) if not f.name.startswith('_')]
    assert ref_fields == our_fields, (ref_fields, our_fields)
    for order in (('nbr', 'neg', 'dedup'), ('dedup', 'nbr', 'neg')):
        a, b = ref_hooks.HookManager(keys=['k']), our_hooks.HookManager(keys=['k'])
        for name in order:
            a.register('k', theirs[name])
            b.register('k', ours[name])
        a.resolve_hooks('k')
        b.resolve_hooks('k')
        assert [type(h).__name__ for h in a._key_to_hooks['k']] == [type(h).__name__ for h in b._key_to_hooks['k']], order
    ok('DGBatch fields and our HookManager\'s resolved order equal the reference\'s')
    # 5. parameter interchange
    from tgm.nn import TGAT as RefTGAT

    from tgm_amd.nn import TGAT

    kw = dict(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2, n_heads=2, dropout=0.1)
    a, b = RefTGAT(**kw).state_dict(), TGAT(**kw).state_dict()
    assert list(a) == list(b) and all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    TGAT(**kw).load_state_dict(a)
    RefTGAT(**kw).load_state_dict(b)
    ok(f'TGAT state_dict: {len(a)} tensors with the reference\'s names / shapes / dtypes, loadable both ways')
    # 6. the reference's unit tests over our classes
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(HERE, 'replay_reference_tests.py')], capture_output=True, text=True)
    tail = [ln for ln in r.stdout.splitlines() if ln.startswith('[replay]')]
    assert r.returncode == 0, '\n'.join(tail) or r.stdout[-2000:] + r.stderr[-2000:]
    ok(tail[0][len('[replay] '):])


if __name__ == '__main__':
    main()
