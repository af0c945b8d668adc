"""The C-ABI library loads without a GPU and exports exactly what include/tgm_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'tgm_amd.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tgmx_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_families():
    syms = declared_symbols()
    for needed in ('tgmx_version', 'tgmx_last_error', 'tgmx_recency_lookup_csr', 'tgmx_ring_lookup', 'tgmx_ring_update',
                   'tgmx_ring_reset', 'tgmx_csr_build', 'tgmx_time2vec', 'tgmx_sgemm_nt', 'tgmx_tgat_attn_reduce'):  # fmt: skip
        assert needed in syms


def test_library_exports_every_declared_symbol():
    from tgm_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared in tgm_amd.h but not exported by libtgm_amd.so'
    # and the ctypes signature table covers the same set
    assert sorted(_native.SIGNATURES) == declared_symbols()
    assert _native.load().tgmx_version() == 7


def test_no_cpu_fallback():
    """Product path fails loudly off-GPU instead of computing on the host."""
    import torch

    from tgm_amd import DGData, DGraph
    from tgm_amd.exceptions import NativeLibraryError
    from tgm_amd.hooks import RecencyNeighborHook
    from tgm_amd.nn import TGAT, Time2Vec

    ei = torch.IntTensor([[0, 1], [0, 2], [2, 3], [2, 0]])
    dg = DGraph(DGData.from_raw(torch.LongTensor([1, 2, 3, 4]), ei, torch.rand(4, 2)))
    hook = RecencyNeighborHook(4, [1], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time'])
    with pytest.raises(NativeLibraryError):
        hook(dg, dg.materialize())
    with pytest.raises(NativeLibraryError):
        Time2Vec(4)(torch.arange(3))
    enc = TGAT(node_dim=1, edge_dim=2, time_dim=4, embed_dim=4, num_layers=1).eval()
    with pytest.raises(NativeLibraryError):
        enc(torch.rand(4, 1), [torch.zeros(2, dtype=torch.int32)], [torch.zeros(2, dtype=torch.int64)],
            [torch.zeros(2, 1, dtype=torch.int32)], [torch.zeros(2, 1, 2)], [torch.zeros(2, 1, dtype=torch.int64)])  # fmt: skip


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'tgm_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f'{f} imports oracle/'


def test_struct_mirrors_have_the_library_s_sizes():
    """The ctypes mirrors of the argument structs must match what the library was compiled with."""
    import ctypes

    from tgm_amd import _native

    lib = _native.load()
    mirrors = {1: _native.RecencyStep, 2: _native.TgatLayer, 3: _native.TgatModel, 4: _native.TgatHop, 5: _native.TgatLayout,
               6: _native.Pipeline, 7: _native.PipelineOut, 8: _native.Dropout, 9: _native.TgnMemoryFwd,
               10: _native.TconvFwd, 11: _native.PipelinePost, 12: _native.TgnStep}
    assert lib.tgmx_abi_sizeof(0) == 16
    for which, cls in mirrors.items():
        assert lib.tgmx_abi_sizeof(which) == ctypes.sizeof(cls), cls.__name__
