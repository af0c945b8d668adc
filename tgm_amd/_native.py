"""ctypes binding of ``libtgm_amd.so`` (C ABI: ``include/tgm_amd.h``).

The library is built in-tree by ``__graft_entry__.build()`` /
``tgm_amd/csrc/Makefile``.  There is deliberately NO fallback: if the shared
object is missing, or no ROCm device is visible, every compute entry point
raises :class:`NativeLibraryError`.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

from .exceptions import NativeLibraryError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libtgm_amd.so')

# every symbol include/tgm_amd.h declares: name -> (restype, argtypes)
_P = c_void_p
SIGNATURES = {
    'tgmx_version': (c_int32, []),
    'tgmx_abi_sizeof': (c_size_t, [c_int32]),
    'tgmx_last_error': (c_char_p, []),
    'tgmx_event_create': (c_int32, [ctypes.POINTER(c_void_p)]),
    'tgmx_event_destroy': (c_int32, [_P]),
    'tgmx_event_elapsed_ms': (c_int32, [_P, _P, ctypes.POINTER(ctypes.c_float)]),
    'tgmx_recency_lookup_csr': (
        c_int32,
        [_P, _P, _P, c_int32, _P, _P, c_int64, c_int32, c_int32, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P],
    ),
    'tgmx_ring_lookup': (
        c_int32,
        [_P, _P, _P, c_int32, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P],
    ),
    'tgmx_ring_update': (
        c_int32,
        [_P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P],
    ),
    'tgmx_ring_reset': (c_int32, [_P, _P, c_int32, c_int32, _P]),
    'tgmx_time2vec': (c_int32, [_P, c_int32, _P, _P, c_int32, c_int64, _P, _P]),
    'tgmx_gather_rows': (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, c_int64, _P]),
    'tgmx_tgat_rres': (c_int32, [_P, c_int64, c_int32, _P, _P, c_int32, c_int32, c_int64, _P, c_int64, _P]),
    'tgmx_sgemm_nt': (
        c_int32,
        [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int32, c_int32, c_int64, c_int64, c_int64, _P],
    ),
    'tgmx_tgat_attn_reduce': (
        c_int32,
        [_P, _P, c_int32, _P, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, ctypes.c_float, c_int32, _P, _P, _P, _P],
    ),
    'tgmx_tgn_store': (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int64, _P, _P, _P, _P, _P, _P]),
    'tgmx_tgn_aggregate': (
        c_int32,
        [_P, c_int64, _P, _P, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, _P, c_int32, c_int32, _P, _P, _P, c_int64, _P],
    ),
    'tgmx_tgn_commit_assoc': (c_int32, [_P, _P, c_int64, _P, c_int64, _P, _P, c_int32, c_int32, _P, _P, _P, _P]),
    'tgmx_tgn_store_batch': (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    'tgmx_tgn_edge_list': (c_int32, [_P, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    'tgmx_tgn_compact_workspace_bytes': (c_size_t, [c_int32]),
    'tgmx_tgn_compact': (c_int32, [_P, _P, _P, _P, c_int32, _P, _P, _P, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    'tgmx_tgn_edge_list_by_id': (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    'tgmx_tgn_gru_gate': (c_int32, [_P, _P, _P, c_int32, c_int64, _P, _P]),
    'tgmx_tgn_commit': (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P, _P]),
    'tgmx_tconv_edge_attr': (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P]),
    'tgmx_tconv_attend': (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, ctypes.c_float, _P, _P, _P]),
    'tgmx_gcn_norm_dense': (c_int32, [_P, _P, _P, c_int64, c_int64, ctypes.c_float, c_int32, _P, c_int64, _P, _P]),
    'tgmx_tgcn_concat': (c_int32, [_P, c_int64, _P, _P, c_int32, c_int64, _P, _P]),
    'tgmx_tgcn_output': (c_int32, [_P, _P, _P, c_int64, _P, _P]),
    'tgmx_tgcn_forward': (c_int32, [_P, _P]),
    'tgmx_tgcn_gate_backward': (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P]),
    'tgmx_tgcn_reset_backward': (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int64, _P, _P, _P]),
    'tgmx_sgemm_tn_workspace_bytes': (c_size_t, [c_int64, c_int32, c_int32, c_int32]),
    'tgmx_sgemm_tn': (
        c_int32,
        [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int32, _P, _P],
    ),
    'tgmx_colsum': (c_int32, [_P, c_int64, c_int64, c_int32, _P, c_int32, _P, _P]),
    'tgmx_relu_mask': (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, _P]),
    'tgmx_add_cols': (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P]),
    'tgmx_ln_backward': (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int32, ctypes.c_float, c_int64, _P, c_int64, _P, c_int64, _P]),
    'tgmx_tgat_attn_backward': (
        c_int32,
        [_P, _P, _P, _P, c_int32, _P, c_int32, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, ctypes.c_float, c_int32, _P, _P, _P, _P, _P],
    ),
    'tgmx_dropout': (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, c_int64, _P]),
    'tgmx_ln_residual_concat': (c_int32, [_P, c_int64, _P, c_int64, _P, _P, c_int32, ctypes.c_float, _P, c_int32, c_int64, _P, c_int64, _P]),
}

TGAT_MAX_LAYERS = 4
E_UNSUPPORTED = -3  # TGMX_E_UNSUPPORTED


class Dropout(ctypes.Structure):
    """tgmx_dropout_t (include/tgm_amd.h)."""

    _fields_ = [('p', ctypes.c_float), ('seed', ctypes.c_uint64), ('stream', ctypes.c_uint64), ('row0', c_int64)]


def dropout_desc(p: float, seed: int, stream: int, row0: int = 0):
    """byref(tgmx_dropout_t) for an active dropout site, None (NULL) when p == 0."""
    if not p:
        return None
    d = Dropout(float(p), seed & 0xFFFFFFFFFFFFFFFF, stream & 0xFFFFFFFFFFFFFFFF, row0)
    return ctypes.byref(d)


class TgatLayer(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('W_Q', 'W_K_t', 'W_V', 'W_O', 'b_O', 'ln_g', 'ln_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'qf_U', 'qf_v',
                                        'W_V_t16', 'W_O_t16', 'fc1_t16', 'fc2_t16', 'qf_lane', 'W_V_t16c')] + [
        (n, c_int32) for n in ('d', 'D', 'T', 'O', 'H', 'emb', 'emb_out')
    ] + [('ln_eps', ctypes.c_float)]


class TgatModel(ctypes.Structure):
    _fields_ = [('tw', c_void_p), ('tb', c_void_p), ('num_layers', c_int32), ('d0', c_int32), ('layers', TgatLayer * TGAT_MAX_LAYERS), ('drop', Dropout)]


class TgatHop(ctypes.Structure):
    _fields_ = [('seed_t', c_void_p), ('nbr_id', c_void_p), ('nbr_t', c_void_p), ('edge_x', c_void_p), ('k', c_int32), ('seed_keyed', c_int32),
                ('nbr_eid', c_void_p), ('edge_table', c_void_p)]  # fmt: skip


class TgatLayerGrads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('W_Q', 'W_KV', 'W_O', 'b_O', 'ln_g', 'ln_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b')]


class TgatGrads(ctypes.Structure):
    _fields_ = [('tw', c_void_p), ('tb', c_void_p), ('layers', TgatLayerGrads * TGAT_MAX_LAYERS)]


PACK_MAX_JOBS = 32


class PackJob(ctypes.Structure):
    """tgmx_pack_job_t"""
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('src_ld', c_int64), ('dst_ld', c_int64), ('rows', c_int32), ('cols', c_int32),
                ('dst_rows', c_int32), ('dst_cols', c_int32), ('transpose', c_int32), ('reserved_', c_int32)]  # fmt: skip


class TgatLayerLayout(ctypes.Structure):
    _fields_ = [(n, c_int64) for n in ('R', 'rres', 'oattn', 'y', 'Q', 'qf', 'zbar', 'cat', 'h1', 'probs', 'out')] + [
        (n, c_int32) for n in ('Op', 'dhp', 'Cp', 'Kc', 'Ep')
    ]


class TgatLayout(ctypes.Structure):
    _fields_ = [('total_bytes', c_int64), ('z0', c_int64), ('level_rows', c_int64 * (TGAT_MAX_LAYERS + 1)),
                ('level_off', c_int64 * (TGAT_MAX_LAYERS + 2)), ('layers', TgatLayerLayout * TGAT_MAX_LAYERS),
                ('compact', c_int64 * (TGAT_MAX_LAYERS + 1))]  # fmt: skip


MAX_SEED_GROUPS = 8
MAX_HOPS = 8


class RecencyStep(ctypes.Structure):
    """tgmx_recency_step_t (include/tgm_amd.h)."""

    _fields_ = [
        ('ring', c_void_p), ('write_pos', c_void_p), ('ring_x', c_void_p),
        ('indptr', c_void_p), ('ev_lo', c_int64), ('ev_hi', c_int64),
        ('D', c_int32), ('B', c_int32), ('num_nodes', c_int32),
        ('n_groups', c_int32),
        ('grp_nid', c_void_p * MAX_SEED_GROUPS), ('grp_ts', c_void_p * MAX_SEED_GROUPS), ('grp_n', c_int64 * MAX_SEED_GROUPS),
        ('seed_nid0', c_void_p), ('seed_ts0', c_void_p), ('S0', c_int64),
        ('n_hops', c_int32), ('k', c_int32 * MAX_HOPS),
        ('out_nid', c_void_p * MAX_HOPS), ('out_ts', c_void_p * MAX_HOPS), ('out_x', c_void_p * MAX_HOPS),
        ('src', c_void_p), ('dst', c_void_p), ('ts', c_void_p), ('edge_x', c_void_p),
        ('n', c_int64), ('eid0', c_int64), ('directed', c_int32), ('key_wrap32', c_int32),
        ('scratch', c_void_p), ('status', c_void_p),
        ('timed_hop', c_int32), ('ev_start', c_void_p), ('ev_stop', c_void_p),
        ('ts_bound', c_int64),
        ('neg_group', c_int32), ('neg_low', c_int32), ('neg_high', c_int32),
        ('neg_seed', ctypes.c_uint64), ('neg_call', ctypes.c_uint64), ('neg_out', c_void_p), ('neg_time_out', c_void_p),
        ('guard_seed_errors', c_int32), ('sorted_ts', c_int32),
        ('out_valid', c_void_p * MAX_HOPS), ('out_valid_prev', c_void_p * MAX_HOPS),
        ('neg_index0', c_int64), ('csr_cursor', c_void_p), ('csr_x_by_pos', c_int32), ('out_eid', c_void_p * MAX_HOPS),
    ]  # fmt: skip


class PipelinePost(ctypes.Structure):
    """tgmx_pipeline_post_t (include/tgm_amd.h)."""

    _fields_ = [
        ('dedup', c_int32), ('dedup_neg', c_int32), ('dedup_nbr', c_int32), ('num_nodes', c_int32), ('dedup_ws', c_void_p), ('uniq_out', c_void_p),
        ('edge_hop', c_int32), ('edge_cap', c_int64), ('row_off', c_void_p), ('edge_index', c_void_p), ('edge_t', c_void_p), ('edge_x', c_void_p),
        ('dev_sizes', c_void_p), ('host_sizes', c_void_p), ('sizes_ready', c_void_p),
    ]  # fmt: skip


class TgnMemoryFwd(ctypes.Structure):
    """tgmx_tgn_memory_fwd_t (include/tgm_amd.h)."""

    _fields_ = [
        ('nodes', c_void_p), ('R', c_int64), ('memory', c_void_p), ('last_update', c_void_p), ('M', c_int32), ('num_nodes', c_int32),
        ('st_lo_s', c_void_p), ('st_cnt_s', c_void_p), ('st_lo_d', c_void_p), ('st_cnt_d', c_void_p),
        ('log_other', c_void_p), ('log_t', c_void_p), ('log_raw', c_void_p), ('D', c_int32),
        ('tw', c_void_p), ('tb', c_void_p), ('T', c_int32), ('mean', c_int32),
        ('W_ih', c_void_p), ('b_ih', c_void_p), ('W_hh', c_void_p), ('b_hh', c_void_p),
        ('ws_aggr', c_void_p), ('ws_h', c_void_p), ('ws_gi', c_void_p), ('ws_gh', c_void_p),
        ('out_mem', c_void_p), ('out_lu', c_void_p), ('assoc', c_void_p), ('stamp', c_int64),
    ]  # fmt: skip


class TgnStep(ctypes.Structure):
    """tgmx_tgn_step_t (include/tgm_amd.h)."""

    _fields_ = [
        ('mem', c_void_p), ('conv', c_void_p),
        ('src', c_void_p), ('dst', c_void_p), ('t', c_void_p), ('raw', c_void_p), ('n', c_int32),
        ('memory', c_void_p), ('last_update', c_void_p), ('reuse_status', c_void_p),
        ('log_base', c_int64), ('log_other', c_void_p), ('log_t', c_void_p), ('log_raw', c_void_p),
        ('st_lo_s', c_void_p), ('st_cnt_s', c_void_p), ('st_lo_d', c_void_p), ('st_cnt_d', c_void_p),
    ]  # fmt: skip


class TgcnFwd(ctypes.Structure):
    """tgmx_tgcn_fwd_t (include/tgm_amd.h)."""

    _fields_ = [
        ('x', c_void_p), ('N', c_int64), ('in_ch', c_int32), ('C', c_int32),
        ('src', c_void_p), ('dst', c_void_p), ('edge_w', c_void_p), ('E', c_int64), ('fill', ctypes.c_float), ('add_self_loops', c_int32),
        ('W3', c_void_p), ('b3', c_void_p),
        ('lin_w', c_void_p * 3), ('lin_b', c_void_p * 3),
        ('H', c_void_p),
        ('A', c_void_p), ('ldA', c_int64), ('norm_ws', c_void_p), ('xwt', c_void_p), ('G', c_void_p), ('cat', c_void_p), ('pre', c_void_p * 3),
        ('out', c_void_p), ('idx32', c_int32),
    ]  # fmt: skip


class TconvFwd(ctypes.Structure):
    """tgmx_tconv_fwd_t (include/tgm_amd.h)."""

    _fields_ = [
        ('x', c_void_p), ('U', c_int64), ('in_ch', c_int32), ('last_update_local', c_void_p),
        ('src', c_void_p), ('tgt', c_void_p), ('t', c_void_p), ('msg', c_void_p), ('E', c_int64), ('D', c_int32), ('T', c_int32),
        ('tw', c_void_p), ('tb', c_void_p), ('W4', c_void_p), ('b4', c_void_p), ('W_edge', c_void_p), ('H', c_int32), ('C', c_int32),
        ('edge_attr', c_void_p), ('qkvs', c_void_p), ('eproj', c_void_p), ('order', c_void_p), ('seg_lo', c_void_p), ('seg_hi', c_void_p),
        ('sort_ws', c_void_p), ('sort_ws_bytes', c_size_t), ('status', c_void_p),
        ('tgt_count', c_void_p), ('cursor', c_void_p), ('order_big', c_void_p),
    ]  # fmt: skip


SEED_SRC, SEED_DST, SEED_NEG = 0, 1, 2


class Pipeline(ctypes.Structure):
    """tgmx_pipeline_t (include/tgm_amd.h)."""

    _fields_ = [
        ('src', c_void_p), ('dst', c_void_p), ('ts', c_void_p), ('edge_x', c_void_p), ('num_edges', c_int64),
        ('rank', c_int32), ('world', c_int32), ('n_roles', c_int32), ('seed_role', c_int32 * MAX_SEED_GROUPS),
        ('neg_low', c_int32), ('neg_high', c_int32), ('neg_seed', ctypes.c_uint64),
        ('update', c_int32), ('reserved0', c_int32), ('step', RecencyStep),
    ]  # fmt: skip


class PipelineOut(ctypes.Structure):
    """tgmx_pipeline_out_t (include/tgm_amd.h)."""

    _fields_ = [
        ('neg', c_void_p), ('neg_time', c_void_p), ('seed_nid0', c_void_p), ('seed_ts0', c_void_p),
        ('out_nid', c_void_p * MAX_HOPS), ('out_ts', c_void_p * MAX_HOPS), ('out_x', c_void_p * MAX_HOPS),
        ('timed_hop', c_int32), ('ev_start', c_void_p), ('ev_stop', c_void_p),
        ('out_valid', c_void_p * MAX_HOPS), ('out_valid_prev', c_void_p * MAX_HOPS), ('out_eid', c_void_p * MAX_HOPS),
    ]  # fmt: skip


ACCOUNTING_PARTIALS = 1024
SIGNATURES['tgmx_lookup_accounting'] = (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P])
SIGNATURES['tgmx_uniform_lookup_csr'] = (c_int32, [_P, _P, _P, c_int32, _P, c_int64, c_int32, c_int64, c_int32, c_int32, ctypes.c_uint64, ctypes.c_uint64,
                                                   _P, _P, _P, _P, _P])
SIGNATURES['tgmx_ring_update_scratch_bytes'] = (c_size_t, [c_int64, c_int32])
SIGNATURES['tgmx_tgn_gru_gate_backward'] = (c_int32, [_P, _P, _P, _P, c_int32, c_int64, _P, _P, _P])
SIGNATURES['tgmx_tgn_aggregate_backward'] = (c_int32, [c_int64, _P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, c_int32, c_int32, _P, _P, _P])
SIGNATURES['tgmx_tconv_edge_attr_backward'] = (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, _P, _P])
SIGNATURES['tgmx_tconv_attend_backward'] = (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P])
SIGNATURES['tgmx_group_ids'] = (c_int32, [_P, c_int32, _P, _P, _P, _P, _P, _P])
SIGNATURES['tgmx_group_ids_workspace_bytes'] = (c_size_t, [c_int64])
SIGNATURES['tgmx_group_ids_large'] = (c_int32, [_P, c_int64, _P, _P, _P, _P, _P, _P, c_size_t, _P])
SIGNATURES['tgmx_random_negatives'] = (c_int32, [c_int32, c_int32, c_int64, ctypes.c_uint64, ctypes.c_uint64, _P, _P, c_int64, _P, _P])
SIGNATURES['tgmx_random_negatives_at'] = (c_int32, [c_int32, c_int32, c_int64, ctypes.c_uint64, ctypes.c_uint64, c_int64, _P, _P, c_int64, _P, _P])
SIGNATURES['tgmx_unique_ids_workspace_bytes'] = (c_size_t, [c_int32])
SIGNATURES['tgmx_unique_ids'] = (c_int32, [_P, _P, c_int32, c_int32, _P, _P, _P, _P, _P])
SIGNATURES['tgmx_csr_build_workspace_bytes'] = (c_size_t, [c_int64, c_int32, c_int32])
SIGNATURES['tgmx_csr_build'] = (c_int32, [_P, _P, _P, c_int64, c_int32, _P, c_int64, c_int32, _P, _P, _P, c_size_t, _P, _P])
SIGNATURES['tgmx_recency_step'] = (c_int32, [ctypes.POINTER(RecencyStep), _P])
SIGNATURES['tgmx_recency_step_plan'] = (c_int32, [ctypes.POINTER(RecencyStep)])
SIGNATURES['tgmx_pipeline_step'] = (c_int32, [ctypes.POINTER(Pipeline), c_int64, c_int64, ctypes.c_uint64, ctypes.POINTER(PipelineOut), ctypes.POINTER(PipelinePost), _P])
SIGNATURES['tgmx_event_synchronize'] = (c_int32, [_P])
SIGNATURES['tgmx_event_create_sync'] = (c_int32, [ctypes.POINTER(c_void_p)])
SIGNATURES['tgmx_event_record'] = (c_int32, [_P, _P])
SIGNATURES['tgmx_stream_wait_event'] = (c_int32, [_P, _P])
SIGNATURES['tgmx_stream_handoff'] = (c_int32, [_P, _P, _P])
SIGNATURES['tgmx_worker_create'] = (c_int32, [ctypes.POINTER(c_void_p)])
SIGNATURES['tgmx_worker_destroy'] = (c_int32, [_P])
SIGNATURES['tgmx_worker_pipeline_step'] = (c_int32, [_P, ctypes.POINTER(Pipeline), c_int64, c_int64, ctypes.c_uint64, ctypes.POINTER(PipelineOut),
                                                     ctypes.POINTER(PipelinePost), _P, _P, _P, ctypes.POINTER(ctypes.c_uint64)])
SIGNATURES['tgmx_worker_wait'] = (c_int32, [_P, ctypes.c_uint64])
SIGNATURES['tgmx_slice'] = (c_int32, [_P, c_int64, c_int32, c_int64, c_int32, c_int64, c_int64, c_int64, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)])
SIGNATURES['tgmx_discretize_workspace_bytes'] = (c_size_t, [c_int64])
SIGNATURES['tgmx_discretize_keep'] = (c_int32, [_P, _P, _P, c_int64, ctypes.c_double, _P, _P, _P, _P, c_size_t, _P])
SIGNATURES['tgmx_tgn_memory_forward'] = (c_int32, [ctypes.POINTER(TgnMemoryFwd), _P])
SIGNATURES['tgmx_tconv_forward'] = (c_int32, [ctypes.POINTER(TconvFwd), _P])
SIGNATURES['tgmx_tgn_step'] = (c_int32, [ctypes.POINTER(TgnStep), _P])
SIGNATURES['tgmx_segment_sort_workspace_bytes'] = (c_size_t, [c_int64])
SIGNATURES['tgmx_segment_sort'] = (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P, c_size_t, _P, _P])
SIGNATURES['tgmx_tgat_tile16_floats'] = (c_size_t, [c_int32, c_int32])
SIGNATURES['tgmx_tgat_tile16'] = (c_int32, [_P, c_int64, c_int32, c_int32, _P, _P])
SIGNATURES['tgmx_pack2d'] = (c_int32, [ctypes.POINTER(PackJob), c_int32, _P])

SIGNATURES['tgmx_pair_dedup_workspace_bytes'] = (c_size_t, [c_int64])
SIGNATURES['tgmx_pair_dedup'] = (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, _P, c_size_t, _P])
SIGNATURES['tgmx_tgat_layout'] = (c_int32, [ctypes.POINTER(TgatModel), c_int64, ctypes.POINTER(TgatHop), c_int32, ctypes.POINTER(TgatLayout)])
SIGNATURES['tgmx_tgat_workspace_bytes'] = (c_size_t, [ctypes.POINTER(TgatModel), c_int64, ctypes.POINTER(TgatHop)])
SIGNATURES['tgmx_tgat_backward_workspace_bytes'] = (c_size_t, [ctypes.POINTER(TgatModel), ctypes.POINTER(TgatLayout), ctypes.POINTER(TgatHop)])
SIGNATURES['tgmx_tgat_backward'] = (c_int32, [ctypes.POINTER(TgatModel), ctypes.POINTER(TgatLayout), ctypes.POINTER(TgatHop), _P, _P, c_int64,
                                              ctypes.POINTER(Dropout), ctypes.POINTER(TgatGrads), _P, c_size_t, _P])  # fmt: skip
SIGNATURES['tgmx_tgat_forward'] = (
    c_int32,
    [ctypes.POINTER(TgatModel), _P, c_int64, _P, c_int64, ctypes.POINTER(TgatHop), _P, c_size_t, c_int32, _P, _P],
)

_lib: Optional[ctypes.CDLL] = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen the kernel library and type every exported entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise NativeLibraryError(
            f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C tgm_amd/csrc`). tgm_amd has no CPU fallback.'
        )
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeLibraryError(f'failed to load {path}: {e}') from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f'{path} does not export {name}; rebuild it') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def require_device(t: torch.Tensor, what: str) -> None:
    if t.device.type != 'cuda':
        raise NativeLibraryError(
            f'{what} lives on {t.device}; tgm_amd kernels run only on a ROCm device (there is no CPU fallback). '
            "Create the DGraph with device='cuda'."
        )


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tgmx_last_error()
        raise RuntimeError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cuda_get_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr(device_index: Optional[int] = None) -> int:
    """hipStream_t of torch's current stream (on ``device_index``, default: the current device).  Goes straight to the C
    accessor: ``torch.cuda.current_stream()`` builds a Stream object through several Python layers (~8 us per call)."""
    if _raw_stream is not None:
        if device_index is None:
            device_index = _cuda_get_device() if _cuda_get_device is not None else torch.cuda.current_device()
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


class KernelTimer:
    """A (start, stop) pair of HIP events recorded by the library right around one kernel launch."""

    def __init__(self) -> None:
        lib = load()
        a, b = c_void_p(), c_void_p()
        check(lib.tgmx_event_create(ctypes.byref(a)), 'tgmx_event_create')
        check(lib.tgmx_event_create(ctypes.byref(b)), 'tgmx_event_create')
        self.start, self.stop = a.value, b.value

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        check(load().tgmx_event_elapsed_ms(self.start, self.stop, ctypes.byref(ms)), 'tgmx_event_elapsed_ms')
        return float(ms.value)

    def __del__(self) -> None:
        try:
            lib = load()
            lib.tgmx_event_destroy(self.start)
            lib.tgmx_event_destroy(self.stop)
        except Exception:
            pass
