"""TGN memory module + graph-attention embedding (tgm/nn/encoder/tgn.py) on HIP kernels.

Same classes, constructor arguments, parameter / buffer names (``time_enc``, ``memory_updater``,
``memory``, ``last_update``; ``conv.lin_{key,query,value,edge,skip}``) and call protocol as the
reference (``memory(n_id)``, ``memory.update_state(src, dst, t, raw_msg)``, ``reset_state``,
``detach``, ``train``/``eval`` with the flush of tgn.py:245-251).  Forward arithmetic only.

Execution model: the reference's per-node Python dict of stored events is an append-only device
log + a (lo, cnt) window per node and role; commits over a batch's touched nodes run on all 2*bs
batch entries with a first-occurrence flag (no ``unique``, no host synchronisation); the GRU
contractions run on the exact-fp32 MFMA GEMM.  ``torch.sort`` / ``searchsorted`` are used as
device-side glue to group a batch's entries by node.

Tie note: the reference orders a node's events inside one batch with a NON-stable sort
(tgn.py:226), so which of two events with the same float32 timestamp ``LastAggregator`` picks is
unspecified there; here it is the earlier edge of the batch.
"""
from __future__ import annotations

import copy
import ctypes
import os
from typing import Callable, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import _native
from . import _ops
from ._paramver import TransientCaches, param_key, param_list
from .time_encoding import Time2Vec

_COMPOSE_IN_PYTHON = bool(os.environ.get('TGMX_TGN_PY'))  # A/B knob: module forwards as sequences of ctypes calls instead of one C driver call


class LastAggregator(nn.Module):
    """Keep, per node, the message with the largest timestamp (tgn.py:43-56)."""

    mean = 0


class MeanAggregator(nn.Module):
    """Average a node's messages (tgn.py:59-63)."""

    mean = 1


class IdentityMessage(nn.Module):
    """message = [mem[src] | mem[dst] | raw_msg | t_enc]  (tgn.py:66-74)."""

    def __init__(self, raw_msg_dim: int, memory_dim: int, time_dim: int) -> None:
        super().__init__()
        self.out_channels = raw_msg_dim + 2 * memory_dim + time_dim


class TGNMemory(TransientCaches, nn.Module):
    _TRANSIENT = ('_fwd_args', '_group_ws', '_fwd')

    def __init__(self, num_nodes: int, raw_msg_dim: int, memory_dim: int, time_dim: int, message_module: Callable,
                 aggregator_module: Callable) -> None:  # fmt: skip
        super().__init__()
        if not isinstance(message_module, IdentityMessage):
            raise NotImplementedError('tgm_amd TGNMemory supports IdentityMessage only')
        if not isinstance(aggregator_module, (LastAggregator, MeanAggregator)):
            raise NotImplementedError('tgm_amd TGNMemory supports LastAggregator / MeanAggregator only')
        self.num_nodes, self.raw_msg_dim, self.memory_dim, self.time_dim = num_nodes, raw_msg_dim, memory_dim, time_dim
        self.msg_s_module = message_module
        self.msg_d_module = copy.deepcopy(message_module)
        self.aggr_module = aggregator_module
        self.time_enc = Time2Vec(time_dim=time_dim)
        self.memory_updater = nn.GRUCell(message_module.out_channels, memory_dim)
        self.register_buffer('memory', torch.zeros(num_nodes, memory_dim))
        self.register_buffer('last_update', torch.zeros(num_nodes, dtype=torch.long))
        self.register_buffer('_assoc', torch.empty(num_nodes, dtype=torch.long))
        # message store: per role a (lo, cnt) window per node into the shared event log
        self._st_lo = [None, None]
        self._st_cnt = [None, None]
        self._log_other: Optional[Tensor] = None
        self._log_t: Optional[Tensor] = None
        self._log_raw: Optional[Tensor] = None
        self._log_len = 0
        self._log_cap_min = 1 << 16  # rows; tests shrink it to exercise the compaction
        # reuse_forward (ours, default off = the reference's evaluation order): in train mode update_state commits the rows the
        # preceding forward() just produced for the batch's nodes instead of recomputing them (same state, same arithmetic,
        # same values; tgn.py:165-177 calls _get_updated_memory a second time).  Contract: the batch's src / dst nodes are
        # among that forward's n_id (the reference loop passes batch.unique_nids) -- verified on the device, reported by check().
        self.reuse_forward = False
        self._version = 0          # bumped by every mutation of memory / last_update / the message stores
        self._fwd = None           # (version, stamp, memory rows [R, M], last_update rows [R]) of the last train-mode forward
        self._assoc64 = None       # [N] int64: (stamp << 32) | row
        self._stamp = 0
        self._reuse_status = None  # device int32: 1 = an update_state node was not part of the reused forward
        self._fwd_args = None      # tgmx_tgn_memory_fwd_t, reused
        self.shard_commits = True  # under torch.distributed (world > 1): shard update_state's commit across ranks
        # a list (ours, None = off): every sharded commit appends (start event, stop event, commit rows, bytes sent, bytes received)
        # around its all-gather -- HIP events on the stream the collective is issued from (bench.py's tgn_memory_allgather block)
        self.allgather_log: Optional[list] = None
        self.memory_updater.reset_parameters()

    # -- state ------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self.time_enc.w.weight.device

    def reset_parameters(self) -> None:
        self.memory_updater.reset_parameters()
        self.reset_state()

    def reset_state(self) -> None:
        self._version += 1
        self.memory.zero_()
        self.last_update.zero_()
        self._reset_message_store()

    def detach(self) -> None:
        self.memory.detach_()

    def _reset_message_store(self) -> None:
        self._version += 1
        dev = self.memory.device
        for r in (0, 1):
            self._st_lo[r] = torch.zeros(self.num_nodes, dtype=torch.int64, device=dev)
            self._st_cnt[r] = torch.zeros(self.num_nodes, dtype=torch.int32, device=dev)
        self._log_len = 0

    def _ensure_store(self, extra: int) -> None:
        """Room for ``extra`` more log rows.  The log is append-only between compactions; a node's window is dead as soon
        as the node shows up again in that role (``_update_msg_store`` REPLACES a node's events, tgn.py:218-229), so when
        the log is full the live windows are copied -- packed -- into FRESH tensors (autograd contexts that snapshotted
        windows of the old tensors keep them alive and stay valid) and the capacity only grows when the live rows need it.
        Memory is bounded by the live windows (what the reference's per-node store holds), not by the events seen."""
        dev = self.memory.device
        if self._st_lo[0] is None or self._st_lo[0].device != dev:
            self._reset_message_store()
            self._log_other = None
        need = self._log_len + extra
        cap = 0 if self._log_other is None else self._log_other.numel()
        if need > cap or (self._log_other is not None and self._log_other.device != dev):
            live = 0
            if self._log_len:
                live = int(self._st_cnt[0].sum()) + int(self._st_cnt[1].sum())  # one host read per compaction (rare)
            new_cap = max(2 * (live + extra), self._log_cap_min)
            self._compact_into(new_cap, dev, live)

    def _compact_into(self, new_cap: int, dev: torch.device, live: Optional[int] = None) -> None:
        """Pack every live window into new log tensors of ``new_cap`` rows and re-point the windows."""
        other = torch.empty(new_cap, dtype=torch.int32, device=dev)
        t = torch.empty(new_cap, dtype=torch.int64, device=dev)
        raw = torch.empty((new_cap, max(self.raw_msg_dim, 1)), dtype=torch.float32, device=dev)
        base = 0
        if self._log_len:
            # one scan of the 2 N counts + one move launch (tgmx_tgn_compact); the total is the sum the caller sized new_cap from
            lib = _native.load()
            base = live if live is not None else int(self._st_cnt[0].sum()) + int(self._st_cnt[1].sum())
            if base > new_cap:
                raise RuntimeError(f'TGNMemory: {base} live message rows do not fit the new log of {new_cap}')
            need = int(lib.tgmx_tgn_compact_workspace_bytes(self.num_nodes))
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            _native.check(
                lib.tgmx_tgn_compact(self._st_lo[0].data_ptr(), self._st_cnt[0].data_ptr(), self._st_lo[1].data_ptr(), self._st_cnt[1].data_ptr(),
                                     self.num_nodes, self._log_other.data_ptr(), self._log_t.data_ptr(), self._log_raw.data_ptr(), self.raw_msg_dim,
                                     other.data_ptr(), t.data_ptr(), raw.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr()),
                'tgmx_tgn_compact',
            )  # fmt: skip
        self._log_other, self._log_t, self._log_raw = other, t, raw
        self._log_len = base

    # -- kernels ----------------------------------------------------------------
    def _updated(self, nodes: Tensor, record: bool = False) -> Tuple[Tensor, Tensor]:
        """Look-ahead memory for int32 node ids ``nodes`` [R] (tgn.py:191-216); writes nothing (``record``: also node -> row
        in the private association table, for ``reuse_forward``)."""
        lib = _native.load()
        _native.require_device(nodes, 'n_id')
        self._ensure_store(0)
        assoc, stamp = None, 0
        if record:
            if self._assoc64 is None or self._assoc64.device != nodes.device:
                self._assoc64 = torch.zeros(self.num_nodes, dtype=torch.int64, device=nodes.device)
                self._reuse_status = torch.zeros(1, dtype=torch.int32, device=nodes.device)
            self._stamp += 1
            assoc, stamp = self._assoc64, self._stamp
        if torch.is_grad_enabled() and any(p.requires_grad for p in param_list(self)):
            return self._updated_train(nodes, assoc, stamp)
        dev, R, M, D, T = nodes.device, nodes.numel(), self.memory_dim, self.raw_msg_dim, self.time_dim
        W = 2 * M + D + T
        if _COMPOSE_IN_PYTHON:  # A/B knob: the same launches, one ctypes call each
            stream = _native.stream_ptr()
            aggr = torch.empty((R, W), dtype=torch.float32, device=dev)
            new_lu = torch.empty(R, dtype=torch.int64, device=dev)
            tw, tb = self.time_enc.w.weight.detach().reshape(-1), self.time_enc.w.bias.detach()
            _native.check(
                lib.tgmx_tgn_aggregate(
                    nodes.data_ptr(), R, self.memory.data_ptr(), self.last_update.data_ptr(), M, self.num_nodes,
                    self._st_lo[0].data_ptr(), self._st_cnt[0].data_ptr(), self._st_lo[1].data_ptr(), self._st_cnt[1].data_ptr(),
                    _native.ptr(self._log_other), _native.ptr(self._log_t), _native.ptr(self._log_raw), D, tw.data_ptr(), tb.data_ptr(),
                    T, self.aggr_module.mean, aggr.data_ptr(), new_lu.data_ptr(), _native.ptr(assoc), stamp, stream,
                ),
                'tgmx_tgn_aggregate',
            )  # fmt: skip
            h = _ops.gather_rows(self.memory.detach(), nodes)
            gru = self.memory_updater
            gi = torch.empty((R, 3 * M), dtype=torch.float32, device=dev)
            gh = torch.empty((R, 3 * M), dtype=torch.float32, device=dev)
            _ops.sgemm_nt(aggr, gru.weight_ih.detach(), gi, bias=gru.bias_ih.detach())
            _ops.sgemm_nt(h, gru.weight_hh.detach(), gh, bias=gru.bias_hh.detach())
            out = torch.empty((R, M), dtype=torch.float32, device=dev)
            _native.check(lib.tgmx_tgn_gru_gate(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), M, R, out.data_ptr(), stream), 'tgmx_tgn_gru_gate')
            return out, new_lu
        # one native call (tgmx_tgn_memory_forward): aggregate, gather, the two GRU GEMMs, gates
        ws = torch.empty(R * (W + 7 * M), dtype=torch.float32, device=dev)  # aggr [R, W] | h [R, M] | gi [R, 3M] | gh [R, 3M]
        out = torch.empty((R, M), dtype=torch.float32, device=dev)
        new_lu = torch.empty(R, dtype=torch.int64, device=dev)
        gru = self.memory_updater
        a = self._fwd_args
        if a is None:
            a = self._fwd_args = _native.TgnMemoryFwd()
        tw, tb = self.time_enc.w.weight.detach().reshape(-1), self.time_enc.w.bias.detach()
        base = ws.data_ptr()
        a.nodes, a.R, a.memory, a.last_update, a.M, a.num_nodes = nodes.data_ptr(), R, self.memory.data_ptr(), self.last_update.data_ptr(), M, self.num_nodes
        a.st_lo_s, a.st_cnt_s, a.st_lo_d, a.st_cnt_d = (self._st_lo[0].data_ptr(), self._st_cnt[0].data_ptr(), self._st_lo[1].data_ptr(),
                                                        self._st_cnt[1].data_ptr())  # fmt: skip
        a.log_other, a.log_t, a.log_raw, a.D = _native.ptr(self._log_other), _native.ptr(self._log_t), _native.ptr(self._log_raw), D
        a.tw, a.tb, a.T, a.mean = tw.data_ptr(), tb.data_ptr(), T, self.aggr_module.mean
        a.W_ih, a.b_ih, a.W_hh, a.b_hh = (gru.weight_ih.detach().data_ptr(), gru.bias_ih.detach().data_ptr(), gru.weight_hh.detach().data_ptr(),
                                          gru.bias_hh.detach().data_ptr())  # fmt: skip
        a.ws_aggr, a.ws_h, a.ws_gi, a.ws_gh = base, base + 4 * R * W, base + 4 * R * (W + M), base + 4 * R * (W + 4 * M)
        a.out_mem, a.out_lu, a.assoc, a.stamp = out.data_ptr(), new_lu.data_ptr(), _native.ptr(assoc), stamp
        _native.check(lib.tgmx_tgn_memory_forward(a, _native.stream_ptr()), 'tgmx_tgn_memory_forward')
        return out, new_lu

    def _updated_train(self, nodes: Tensor, assoc: Optional[Tensor] = None, stamp: int = 0) -> Tuple[Tensor, Tensor]:
        """Same arithmetic with autograd through the hand-written backward kernels (nn/_tgn_train.py)."""
        from ._tgn_train import AggregateFn, GruGateFn, LinearFn

        aggr, new_lu = AggregateFn.apply(self.time_enc.w.weight, self.time_enc.w.bias, self, nodes, assoc, stamp)
        h = _ops.gather_rows(self.memory.detach(), nodes)
        gru = self.memory_updater
        gi = LinearFn.apply(aggr, gru.weight_ih, gru.bias_ih)
        gh = LinearFn.apply(h, gru.weight_hh, gru.bias_hh)
        return GruGateFn.apply(gi, gh, h), new_lu

    def check(self) -> None:
        """Raise if an ``update_state`` under ``reuse_forward`` met a node that the reused forward had not evaluated (one
        device -> host read; the offending rows were left unchanged)."""
        if self._reuse_status is not None and int(self._reuse_status.item()):
            self._reuse_status.zero_()
            raise RuntimeError('TGNMemory(reuse_forward=True): update_state saw a node that was not in the preceding forward(n_id)')

    @torch.no_grad()
    def _commit(self, nodes: Tensor, flag: Optional[Tensor]) -> None:
        self._version += 1
        lib = _native.load()
        CH = 1 << 16  # bounds the [rows, msg_dim] scratch when all N nodes are flushed
        # every row is computed from the OLD memory first (the reference evaluates all of n_id at once), then written
        done = []
        if self._sharded(nodes.numel()):
            done.append((0, nodes) + self._updated_sharded(nodes))
        else:
            for lo in range(0, nodes.numel(), CH):
                part = nodes[lo : lo + CH]
                done.append((lo, part) + self._updated(part))
        for lo, part, mem, lu in done:
            fl = None if flag is None else flag[lo : lo + part.numel()]
            _native.check(
                lib.tgmx_tgn_commit(part.data_ptr(), _native.ptr(fl), mem.data_ptr(), lu.data_ptr(), self.memory_dim, self.num_nodes,
                                    part.numel(), self.memory.data_ptr(), self.last_update.data_ptr(), _native.stream_ptr()),
                'tgmx_tgn_commit',
            )  # fmt: skip

    # -- one process per GPU: shard the commit rows, all-gather the updated memory rows ------------
    def _sharded(self, rows: int) -> bool:
        import torch.distributed as dist

        return self.shard_commits and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and rows >= dist.get_world_size()

    def _updated_sharded(self, nodes: Tensor) -> Tuple[Tensor, Tensor]:
        """Each rank evaluates a contiguous slice of the commit rows from its replica of the (identical)
        store and memory; ONE all-gather (RCCL over xGMI when the backend is nccl) of fixed-size
        (memory row, last_update) records gives every replica all rows, applied in the same order
        everywhere, so replicas stay bit-identical.  Payload: rows * (4*M + 8) bytes per step."""
        import torch.distributed as dist

        world, rank = dist.get_world_size(), dist.get_rank()
        R, M, dev = nodes.numel(), self.memory_dim, nodes.device
        per = (R + world - 1) // world
        lo, hi = min(rank * per, R), min((rank + 1) * per, R)
        # one record per row: M floats of memory (padded to an even count) + the int64 last_update as two float-sized words
        Mp = M + (M & 1)
        rec_l = torch.zeros((per, Mp + 2), dtype=torch.float32, device=dev)
        if hi > lo:
            m, l = self._updated(nodes[lo:hi])
            rec_l[: hi - lo, :M] = m
            rec_l[: hi - lo, Mp:].view(torch.int64)[:, 0] = l
        rec_g = torch.empty((world * per, Mp + 2), dtype=torch.float32, device=dev)
        log = self.allgather_log
        if log is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if dist.get_backend() == 'gloo':  # CPU collective (tests): stage through the host
            rc = torch.empty(rec_g.shape)
            dist.all_gather_into_tensor(rc, rec_l.cpu())
            rec_g.copy_(rc)
        else:
            dist.all_gather_into_tensor(rec_g, rec_l)  # the step's only collective
        if log is not None:
            ev1.record()
            log.append((ev0, ev1, R, rec_l.numel() * 4, rec_g.numel() * 4))
        mem_g = rec_g[:, :M].contiguous()
        lu_g = rec_g[:, Mp:].view(torch.int64)[:, 0].contiguous()
        return mem_g[:R], lu_g[:R]

    def _store_batch(self, src32: Tensor, dst32: Tensor, t: Tensor, raw: Optional[Tensor]) -> None:
        """Both roles' message stores of one batch (tgn.py:173,176): one launch for n <= 1024 events."""
        n = src32.numel()
        if n > 1024:
            self._store_role(0, src32, dst32, t, raw)
            self._store_role(1, dst32, src32, t, raw)
            return
        self._version += 1
        self._ensure_store(2 * n)
        _native.check(
            _native.load().tgmx_tgn_store_batch(src32.data_ptr(), dst32.data_ptr(), t.data_ptr(), _native.ptr(raw), self.raw_msg_dim, n, self._log_len,
                                                self._log_other.data_ptr(), self._log_t.data_ptr(), self._log_raw.data_ptr(),
                                                self._st_lo[0].data_ptr(), self._st_cnt[0].data_ptr(), self._st_lo[1].data_ptr(),
                                                self._st_cnt[1].data_ptr(), _native.stream_ptr()),
            'tgmx_tgn_store_batch',
        )  # fmt: skip
        self._log_len += 2 * n

    def _store_role(self, role: int, node: Tensor, other: Tensor, t: Tensor, raw: Optional[Tensor]) -> None:
        self._version += 1
        lib = _native.load()
        n = node.numel()
        self._ensure_store(n)
        if n <= 1024:  # one launch: stable sort by node + run bounds (csrc/tgn.hip group_ids_kernel)
            dev = node.device
            node_sorted = torch.empty(n, dtype=torch.int32, device=dev)
            perm, left, right = (torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3))
            _native.check(lib.tgmx_group_ids(node.data_ptr(), n, node_sorted.data_ptr(), perm.data_ptr(), left.data_ptr(), right.data_ptr(),
                                             None, _native.stream_ptr()), 'tgmx_group_ids')  # fmt: skip
        else:  # any n: one stable radix sort + a finishing launch (tgmx_group_ids_large), no torch ops
            node_sorted, perm, left, right, _ = self._group_large(node, want_first=False)
        _native.check(
            lib.tgmx_tgn_store(perm.data_ptr(), node_sorted.data_ptr(), left.data_ptr(), right.data_ptr(), other.data_ptr(), t.data_ptr(),
                               _native.ptr(raw), self.raw_msg_dim, n, self._log_len, self._log_other.data_ptr(), self._log_t.data_ptr(),
                               self._log_raw.data_ptr(), self._st_lo[role].data_ptr(), self._st_cnt[role].data_ptr(), _native.stream_ptr()),
            'tgmx_tgn_store',
        )  # fmt: skip
        self._log_len += n

    def _group_large(self, ids: Tensor, want_first: bool):
        """(sorted ids, perm, run_lo, run_hi, first-of-run) of more than 1024 int32 ids on our kernels (tgn.py:218-229 sorts per role;
        :165-177 takes the unique endpoints): what torch.sort(stable=True) + 2 x searchsorted / a neighbour compare produced."""
        lib = _native.load()
        n = ids.numel()
        dev = ids.device
        need = int(lib.tgmx_group_ids_workspace_bytes(n))
        ws = getattr(self, '_group_ws', None)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._group_ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=dev)
        srt = torch.empty(n, dtype=torch.int32, device=dev)
        if want_first:
            first = torch.empty(n, dtype=torch.uint8, device=dev)
            perm = left = right = None
        else:
            first = None
            perm, left, right = (torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3))
        _native.check(lib.tgmx_group_ids_large(ids.data_ptr(), n, srt.data_ptr(), _native.ptr(perm), _native.ptr(left), _native.ptr(right),
                                               _native.ptr(first), ws.data_ptr(), ws.numel(), _native.stream_ptr()), 'tgmx_group_ids_large')  # fmt: skip
        return srt, perm, left, right, first

    # -- reference protocol -------------------------------------------------------
    def forward(self, n_id: Tensor) -> Tuple[Tensor, Tensor]:
        """train mode: memory as it would be after committing the stored messages (nothing is written);
        eval mode: the table rows (tgn.py:157-163)."""
        _native.require_device(n_id, 'n_id')
        if self.training:
            if not self.reuse_forward:
                return self._updated(n_id.to(torch.int32).contiguous())
            mem, lu = self._updated(n_id.to(torch.int32).contiguous(), record=True)
            self._fwd = (self._version, self._stamp, mem.detach(), lu, param_key(self))
            return mem, lu
        idx = n_id.long()
        return self.memory[idx], self.last_update[idx]

    def update_state(self, src: Tensor, dst: Tensor, t: Tensor, raw_msg: Tensor) -> None:
        """tgn.py:165-177.  The commit runs over all 2*bs batch entries with a first-occurrence flag."""
        _native.require_device(src, 'src')
        src32, dst32 = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
        t = t.to(torch.int64).contiguous()
        raw = _ops._f32c(raw_msg, 'raw_msg') if self.raw_msg_dim else None
        fwd = self._fwd
        if (self.training and self.reuse_forward and fwd is not None and fwd[0] == self._version and not self._sharded(2 * src32.numel())
                and fwd[4] == param_key(self)):  # (an optimizer step in between: the reference recomputes with the new weights)
            # the rows this batch's nodes need are the ones the forward just computed: commit by row copy, then store
            _, stamp, mem_rows, lu_rows, _ = fwd
            self._version += 1
            _native.check(
                _native.load().tgmx_tgn_commit_assoc(src32.data_ptr(), dst32.data_ptr(), src32.numel(), self._assoc64.data_ptr(), stamp,
                                                     mem_rows.data_ptr(), lu_rows.data_ptr(), self.memory_dim, self.num_nodes,
                                                     self.memory.data_ptr(), self.last_update.data_ptr(), self._reuse_status.data_ptr(),
                                                     _native.stream_ptr()),
                'tgmx_tgn_commit_assoc',
            )  # fmt: skip
            self._fwd = None
            self._store_batch(src32, dst32, t, raw)
            return
        both = torch.cat([src32, dst32])
        if both.numel() <= 1024:
            srt = torch.empty_like(both)
            first = torch.empty(both.numel(), dtype=torch.uint8, device=both.device)
            _native.check(_native.load().tgmx_group_ids(both.data_ptr(), both.numel(), srt.data_ptr(), None, None, None, first.data_ptr(),
                                                        _native.stream_ptr()), 'tgmx_group_ids')  # fmt: skip
        else:
            srt, _, _, _, first = self._group_large(both, want_first=True)
        if self.training:
            self._commit(srt, first)
            self._store_batch(src32, dst32, t, raw)
        else:
            self._store_batch(src32, dst32, t, raw)
            self._commit(srt, first)

    def load_state_dict(self, *args, **kwargs):
        self._version += 1  # memory / last_update may change under a cached forward
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode: bool = True) -> 'TGNMemory':
        if self.training and not mode and self.memory.device.type == 'cuda':
            # entering eval: flush every node's stored messages into the memory, clear the stores (tgn.py:245-251)
            self._commit(torch.arange(self.num_nodes, dtype=torch.int32, device=self.memory.device), None)
            self._reset_message_store()
        super().train(mode)
        return self


def _edge_list_enqueue(batch, hop: int):
    """Enqueue ``tgmx_tgn_edge_list`` for one hop; returns (edge_index [2, cap], edge_t [cap], edge_x [cap, D], count [1] on the device)."""
    lib = _native.load()
    from ..core.lazy import EdgeFeaturesById

    seeds, nbr = batch.seed_nids[hop], batch.nbr_nids[hop]
    nbr_t = batch.nbr_edge_time[hop]
    _native.require_device(nbr, 'nbr_nids')
    dev = nbr.device
    S, k = nbr.shape
    c = lambda t, dt: t if (t.dtype == dt and t.is_contiguous()) else t.to(dt).contiguous()
    seeds, nbr, nbr_t = c(seeds, torch.int32), c(nbr, torch.int32), c(nbr_t, torch.int64)
    by_id = isinstance(batch.nbr_edge_x, EdgeFeaturesById)  # the rows come from the resident store by edge id: no dense copy is materialized
    if by_id:
        eid, table = c(batch.nbr_edge_x.eids[hop], torch.int32), batch.nbr_edge_x.table
        D = table.shape[1]
    else:
        nbr_x = c(batch.nbr_edge_x[hop], torch.float32)
        D = nbr_x.shape[-1]
    pending = batch.__dict__.get('_unique_dev')  # DeduplicationHook's device-side result: ids [capacity] + count, no host size needed
    if pending is not None:
        uniq, U, ucount = pending[0], 0, pending[1].data_ptr()
    else:
        uniq = c(batch.unique_nids, torch.int32)
        U, ucount = uniq.numel(), None
    cap = max(S * k, 1)
    ei = torch.empty((2, cap), dtype=torch.int64, device=dev)
    et = torch.empty(cap, dtype=torch.int64, device=dev)
    ex = torch.empty((cap, D), dtype=torch.float32, device=dev)
    ws = torch.empty(S + 2, dtype=torch.int64, device=dev)  # row offsets [S + 1] | count
    if by_id:
        _native.check(
            lib.tgmx_tgn_edge_list_by_id(seeds.data_ptr(), nbr.data_ptr(), nbr_t.data_ptr(), eid.data_ptr(), table.data_ptr(), S, k, D, uniq.data_ptr(), U,
                                         ucount, cap, ws.data_ptr(), ei.data_ptr(), et.data_ptr(), ex.data_ptr(), ws[S + 1 :].data_ptr(),
                                         _native.stream_ptr()),
            'tgmx_tgn_edge_list_by_id',
        )  # fmt: skip
        return ei, et, ex, ws[S + 1 :]
    _native.check(
        lib.tgmx_tgn_edge_list(seeds.data_ptr(), nbr.data_ptr(), nbr_t.data_ptr(), nbr_x.data_ptr(), S, k, D, uniq.data_ptr(), U, ucount, cap,
                               ws.data_ptr(), ei.data_ptr(), et.data_ptr(), ex.data_ptr(), ws[S + 1 :].data_ptr(), _native.stream_ptr()),
        'tgmx_tgn_edge_list',
    )  # fmt: skip
    return ei, et, ex, ws[S + 1 :]


def sampled_edge_list(batch, hop: int = 0) -> Tuple[Tensor, Tensor, Tensor]:
    """``(edge_index [2, E] int64, edge_time [E] int64, edge_x [E, D] float32)`` of one sampled hop, exactly what the
    reference's TGN loop assembles from a dozen torch ops (examples/linkproppred/tgn.py:80-92):

        mask = nbr != -1;  edge_index = stack([global_to_local(seeds.repeat_interleave(k)[mask]), global_to_local(nbr[mask])])
        edge_time = nbr_edge_time[hop].flatten()[mask];  edge_x = nbr_edge_x[hop].flatten(0, -2)[mask]

    in two launches (``tgmx_tgn_edge_list``) and one device -> host read (E).  Needs ``batch.unique_nids`` (the
    ``DeduplicationHook``) and the sampler's outputs; bit-identical to the torch formulation (tests/test_tgn_gpu.py).
    ``tgm_amd.hooks.SampledEdgeListHook`` is the same as a hook, whose read of E a prefetching loader hides."""
    ei, et, ex, count = _edge_list_enqueue(batch, hop)
    E = int(count.item())
    return ei[:, :E], et[:E], ex[:E]


class TransformerConv(TransientCaches, nn.Module):
    """Graph transformer operator with edge features -- parameters named like
    ``torch_geometric.nn.TransformerConv`` (2.6.1; third-party to the reference, parity unpinned):
    concat=True, root_weight=True, beta=False, bias=True.  In ``.train()`` mode the attention coefficients go through
    dropout after the softmax like PyG's (``F.dropout(alpha, p=self.dropout)``): a counter-based mask per (edge, head),
    seeded from ``torch.initial_seed()``, a fresh stream per forward call, regenerated by the backward."""

    _TRANSIENT = ('_fwd_args', '_seg_ws', '_stacked', '_tgt_count')

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, dropout: float = 0.0, edge_dim: Optional[int] = None) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.heads, self.dropout, self.edge_dim = in_channels, out_channels, heads, dropout, edge_dim
        hc = heads * out_channels
        self.lin_key = nn.Linear(in_channels, hc)
        self.lin_query = nn.Linear(in_channels, hc)
        self.lin_value = nn.Linear(in_channels, hc)
        self.lin_edge = nn.Linear(edge_dim, hc, bias=False)
        self.lin_skip = nn.Linear(in_channels, hc, bias=True)

    def _incoming_segments(self, tgt: Tensor, U: int) -> Tuple[Tensor, Tensor, Tensor]:
        """Edge ids stably grouped by target node and every node's [lo, hi) range in that order (one bounded-bit radix
        sort + a binary search per node: ``tgmx_segment_sort``; the stack it replaces was a 64-bit stable argsort and two
        ``searchsorted`` calls).  Targets outside [0, U) are clamped and flagged in a device status word (no host sync)."""
        lib = _native.load()
        dev, E = tgt.device, tgt.numel()
        if tgt.dtype != torch.int64:
            tgt = tgt.long()
        need = int(lib.tgmx_segment_sort_workspace_bytes(E))
        ws = getattr(self, '_seg_ws', None)
        if ws is None or ws[0].device != dev or ws[0].numel() < need:
            ws = self._seg_ws = (torch.empty(need, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
        order = torch.empty(E, dtype=torch.int64, device=dev)
        seg = torch.empty((2, U), dtype=torch.int64, device=dev)
        _native.check(lib.tgmx_segment_sort(tgt.data_ptr(), E, U, order.data_ptr(), seg[0].data_ptr(), seg[1].data_ptr(), ws[0].data_ptr(),
                                            ws[0].numel(), ws[1].data_ptr(), _native.stream_ptr()), 'tgmx_segment_sort')  # fmt: skip
        return order, seg[0], seg[1]

    def _stacked_projections(self) -> Tuple[Tensor, Tensor]:
        """[4, HC, in] weights and [4, HC] biases of lin_query / lin_key / lin_value / lin_skip, rebuilt only when a
        parameter was reallocated or modified in place (optimizer step, load_state_dict)."""
        lins = (self.lin_query, self.lin_key, self.lin_value, self.lin_skip)
        key = param_key(p for lin in lins for p in (lin.weight, lin.bias))
        cached = getattr(self, '_stacked', None)
        if cached is None or cached[0] != key:
            W4 = torch.stack([lin.weight.detach().float() for lin in lins]).contiguous()
            b4 = torch.stack([lin.bias.detach().float() for lin in lins]).contiguous()
            cached = self._stacked = (key, W4, b4)
        return cached[1], cached[2]

    def _dropout_site(self) -> tuple:
        """(p, seed, stream) of this forward call's attention dropout; p = 0 outside train mode."""
        if not (self.training and self.dropout > 0):
            return (0.0, 0, 0)
        self._drop_calls = getattr(self, '_drop_calls', 0) + 1
        if getattr(self, '_drop_seed', None) is None:
            self._drop_seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        return (float(self.dropout), self._drop_seed, self._drop_calls)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
        lib = _native.load()
        if torch.is_grad_enabled() and (x.requires_grad or (edge_attr is not None and edge_attr.requires_grad) or any(p.requires_grad for p in param_list(self))):
            self._edge_ctx = None
            return self._forward_train(x, edge_index, edge_attr)
        x = _ops._f32c(x, 'x')
        dev, U, H, C = x.device, x.shape[0], self.heads, self.out_channels
        HC = H * C
        f32 = dict(dtype=torch.float32, device=dev)
        # query / key / value / skip projections of the same x: ONE batched launch over the stacked weights
        W4, b4 = self._stacked_projections()
        qkvs = torch.empty((4, U, HC), **f32)
        E = edge_index.shape[1]
        drop = self._dropout_site()
        if drop[0] == 0.0 and getattr(self, '_edge_ctx', None) is not None and E:
            # GraphAttentionEmbedding handed the raw edge inputs over: the whole layer is one native call (tgmx_tconv_forward)
            lu, t, msg, tw, tb, T = self._edge_ctx
            self._edge_ctx = None
            D = msg.shape[1]
            src, tgt = edge_index[0].contiguous(), edge_index[1].contiguous()
            need = int(lib.tgmx_segment_sort_workspace_bytes(E))
            wsd = getattr(self, '_seg_ws', None)
            if wsd is None or wsd[0].device != dev or wsd[0].numel() < need:
                wsd = self._seg_ws = (torch.empty(need, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
            fl = torch.empty(E * (T + D + HC), **f32)  # edge_attr [E, T + D] | eproj [E, HC]
            Ep, Up = E + (E & 1), U + (U & 1)  # (every block 16-byte aligned: the scan kernel moves two int64 per access)
            ints = torch.empty(2 * Ep + 3 * Up, dtype=torch.int64, device=dev)  # order [E] | seg_lo [U] | seg_hi [U] | cursor [U] | order_big [E]
            # per-target edge counts of the counting grouping: zero on entry, left zero by the call -- kept between calls
            cnt = getattr(self, '_tgt_count', None)
            if cnt is None or cnt.device != dev or cnt.numel() < U:
                cnt = self._tgt_count = torch.zeros(max(2 * U, 1 << 14), dtype=torch.int32, device=dev)
            a = getattr(self, '_fwd_args', None)
            if a is None:
                a = self._fwd_args = _native.TconvFwd()
            a.x, a.U, a.in_ch, a.last_update_local = x.data_ptr(), U, x.shape[1], lu.data_ptr()
            a.src, a.tgt, a.t, a.msg, a.E, a.D, a.T = src.data_ptr(), tgt.data_ptr(), t.data_ptr(), msg.data_ptr(), E, D, T
            a.tw, a.tb, a.W4, a.b4, a.W_edge, a.H, a.C = tw.data_ptr(), tb.data_ptr(), W4.data_ptr(), b4.data_ptr(), self.lin_edge.weight.detach().data_ptr(), H, C
            a.edge_attr, a.qkvs, a.eproj = fl.data_ptr(), qkvs.data_ptr(), fl.data_ptr() + 4 * E * (T + D)
            a.order, a.seg_lo, a.seg_hi = ints.data_ptr(), ints.data_ptr() + 8 * Ep, ints.data_ptr() + 8 * (Ep + Up)
            a.sort_ws, a.sort_ws_bytes, a.status = wsd[0].data_ptr(), wsd[0].numel(), wsd[1].data_ptr()
            a.tgt_count, a.cursor, a.order_big = cnt.data_ptr(), ints.data_ptr() + 8 * (Ep + 2 * Up), ints.data_ptr() + 8 * (Ep + 3 * Up)
            _native.check(lib.tgmx_tconv_forward(a, _native.stream_ptr()), 'tgmx_tconv_forward')
            return qkvs[3]
        self._edge_ctx = None
        _ops.sgemm_nt(x, W4[0], qkvs[0], bias=b4, batch=4, sA=0, sB=W4.stride(0), sC=U * HC)
        q, k, v, out = qkvs[0], qkvs[1], qkvs[2], qkvs[3]
        if E:
            eproj = torch.empty((E, HC), **f32)
            _ops.sgemm_nt(_ops._f32c(edge_attr, 'edge_attr'), self.lin_edge.weight.detach(), eproj)
            src, tgt = edge_index[0].contiguous(), edge_index[1].contiguous()  # flow: source_to_target
            order, seg_lo, seg_hi = self._incoming_segments(tgt, U)
            _native.check(
                lib.tgmx_tconv_attend(q.data_ptr(), k.data_ptr(), v.data_ptr(), eproj.data_ptr(), order.data_ptr(), src.data_ptr(),
                                      seg_lo.data_ptr(), seg_hi.data_ptr(), U, H, C, float(C) ** -0.5, out.data_ptr(),
                                      _native.dropout_desc(*drop), _native.stream_ptr()),
                'tgmx_tconv_attend',
            )  # fmt: skip
        return out


def _tconv_forward_train(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
    """TransformerConv forward with autograd (hand-written backward kernels, nn/_tgn_train.py)."""
    from ._tgn_train import LinearFn, TconvAttendFn

    if self.out_channels > 64:
        raise NotImplementedError('tgm_amd TransformerConv backward supports out_channels <= 64')
    x = x.float().contiguous() if (x.dtype != torch.float32 or not x.is_contiguous()) else x
    dev, U, H, C = x.device, x.shape[0], self.heads, self.out_channels
    q = LinearFn.apply(x, self.lin_query.weight, self.lin_query.bias)
    k = LinearFn.apply(x, self.lin_key.weight, self.lin_key.bias)
    v = LinearFn.apply(x, self.lin_value.weight, self.lin_value.bias)
    skip = LinearFn.apply(x, self.lin_skip.weight, self.lin_skip.bias)
    E = edge_index.shape[1]
    if not E:
        return skip
    eproj = LinearFn.apply(edge_attr.float().contiguous(), self.lin_edge.weight, None)
    src, tgt = edge_index[0].contiguous(), edge_index[1].contiguous()
    order, seg_lo, seg_hi = self._incoming_segments(tgt, U)
    return TconvAttendFn.apply(q, k, v, eproj, skip, order, src, seg_lo, seg_hi, H, C, self._dropout_site())


TransformerConv._forward_train = _tconv_forward_train


class GraphAttentionEmbedding(nn.Module):
    """tgn.py:14-40: edge_attr = [Time2Vec(last_update[src] - t) | msg], then TransformerConv(heads=2, dropout=0.1)."""

    def __init__(self, in_channels: int, out_channels: int, msg_dim: int, time_enc: nn.Module) -> None:
        super().__init__()
        self.time_enc = time_enc
        self.conv = TransformerConv(in_channels, out_channels // 2, heads=2, dropout=0.1, edge_dim=msg_dim + time_enc.time_dim)

    def forward(self, x: Tensor, last_update: Tensor, edge_index: Tensor, t: Tensor, msg: Tensor) -> Tensor:
        lib = _native.load()
        _native.require_device(x, 'x')
        E, T = edge_index.shape[1], self.time_enc.time_dim
        msg = _ops._f32c(msg, 'msg')
        D = msg.shape[1]
        edge_index = edge_index.to(torch.int64)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in param_list(self))):
            from ._tgn_train import EdgeAttrFn

            edge_attr = EdgeAttrFn.apply(self.time_enc.w.weight, self.time_enc.w.bias, last_update.to(torch.int64).contiguous(),
                                         edge_index[0].contiguous(), t.to(torch.int64).contiguous(), msg)  # fmt: skip
            return self.conv(x, edge_index, edge_attr)
        tw, tb = self.time_enc.w.weight.detach().reshape(-1), self.time_enc.w.bias.detach()
        lu64, t64 = last_update.to(torch.int64).contiguous(), t.to(torch.int64).contiguous()
        if E and not _COMPOSE_IN_PYTHON and not (self.conv.training and self.conv.dropout > 0):
            # inference: edge encoding + TransformerConv as one native call; the conv picks the raw inputs up
            self.conv._edge_ctx = (lu64, t64, msg, tw, tb, T)
            return self.conv(x, edge_index, None)
        edge_attr = torch.empty((E, T + D), dtype=torch.float32, device=x.device)
        _native.check(
            lib.tgmx_tconv_edge_attr(lu64.data_ptr(), edge_index[0].contiguous().data_ptr(), t64.data_ptr(), msg.data_ptr(), tw.data_ptr(),
                                     tb.data_ptr(), T, D, E, edge_attr.data_ptr(), _native.stream_ptr()),
            'tgmx_tconv_edge_attr',
        )  # fmt: skip
        return self.conv(x, edge_index, edge_attr)


class TGNStep:
    """The model side of one TGN batch as ONE native call (``tgmx_tgn_step``), for the inference / evaluation loop of
    examples/linkproppred/tgn.py:96-116::

        z, last_update = memory(n_id)                                  # train-mode look-ahead (tgn.py:157-163)
        z2 = embedding(z, last_update, edge_index, edge_t, edge_x)     # GraphAttentionEmbedding (tgn.py:14-40)
        memory.update_state(src, dst, t, raw_msg)                      # tgn.py:165-177

    ``step(n_id, edge_index, edge_t, edge_x, src, dst, t, raw_msg)`` returns ``(z2, z, last_update)`` -- the same launches in the same order
    with the same arguments as the three module calls (identical results, ``tests/test_tgn_gpu.py``), issued back to back from C with
    one pass of argument marshalling instead of three: a cfg 3 batch is ~11 launches on the model side and the host pays ~10 us for each
    when every module call is its own Python-mediated sequence.  The fast path needs what the fast paths of the modules need -- no
    autograd, ``memory.training`` with ``memory.reuse_forward``, no attention dropout, 0 < n <= 1024 events, one process (no sharded
    commit) -- and falls back to the three calls otherwise.  ``step.batch(batch)`` takes the attributes of a TGN loader batch
    (``unique_nids``, ``sampled_edge_*`` from ``SampledEdgeListHook``, the batch's edges)."""

    def __init__(self, memory: 'TGNMemory', embedding: 'GraphAttentionEmbedding') -> None:
        self.memory, self.embedding = memory, embedding
        self._args = _native.TgnStep()
        self._ws = self._fl = self._ints = None
        self._wkey = None
        self._static_key = None  # addresses behind the argument blocks' static fields (see __call__)
        self._lib = None
        self._seg_ws_edges, self._seg_ws_bytes = -1, 0  # the segment sort's workspace covers edge lists up to this size
        # PRIVATE argument blocks: the modules rewrite theirs (TGNMemory._fwd_args, TransformerConv._fwd_args) in full on every module call
        # -- a fallback call, or the modules used directly between two steps -- which the static-field cache below could not see
        self._mem_args, self._conv_args = _native.TgnMemoryFwd(), _native.TconvFwd()
        self.fast_calls = self.fallback_calls = 0

    def batch(self, batch):
        return self(batch.unique_nids, batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x, batch.edge_src, batch.edge_dst,
                    batch.edge_time, batch.edge_x)

    def _eligible(self, n_id: Tensor, edge_index: Tensor, src: Tensor, raw_msg: Optional[Tensor]) -> bool:
        mem, conv = self.memory, self.embedding._modules['conv']  # (registered submodule: the dict, not Module.__getattr__ -- see __call__)
        n = src.numel()
        return (not torch.is_grad_enabled() and not _COMPOSE_IN_PYTHON and mem.training and mem.reuse_forward and n_id.is_cuda and n_id.numel() > 0
                and 0 < n <= 1024 and edge_index.shape[1] > 0 and not (conv.training and conv.dropout > 0) and not mem._sharded(2 * n)
                and (raw_msg is not None or mem.raw_msg_dim == 0))

    @staticmethod
    def _grow(buf: Optional[Tensor], numel: int, dtype, dev) -> Tensor:
        if buf is None or buf.numel() < numel or buf.device != dev or buf.dtype != dtype:
            buf = torch.empty(max(numel + numel // 4, 1), dtype=dtype, device=dev)
        return buf

    def __call__(self, n_id: Tensor, edge_index: Tensor, edge_t: Tensor, edge_x: Tensor, src: Tensor, dst: Tensor, t: Tensor,
                 raw_msg: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
        mem, emb = self.memory, self.embedding
        if not self._eligible(n_id, edge_index, src, raw_msg):
            self.fallback_calls += 1
            z, lu = mem(n_id)
            z2 = emb(z, lu, edge_index, edge_t, edge_x)
            mem.update_state(src, dst, t, raw_msg)
            return z2, z, lu
        self.fast_calls += 1
        # Registered buffers / submodules are read from the modules' dicts and plain attributes are written through `__dict__`:
        # nn.Module.__getattr__ / __setattr__ cost 1-2 us per access, a dozen of them per batch on a host-bound step.
        md, bufs = mem.__dict__, mem._buffers
        conv = emb._modules['conv']
        dev = n_id.device
        i32c = lambda v: v if (v.dtype == torch.int32 and v.is_contiguous()) else v.to(torch.int32).contiguous()
        i64c = lambda v: v if (v.dtype == torch.int64 and v.is_contiguous()) else v.to(torch.int64).contiguous()
        f32c = lambda v: v if (v.dtype == torch.float32 and v.is_contiguous()) else v.float().contiguous()
        nodes, src32, dst32, t64 = i32c(n_id), i32c(src), i32c(dst), i64c(t)
        raw = f32c(raw_msg) if mem.raw_msg_dim else None
        n, R = src32.numel(), nodes.numel()
        M, D, T = mem.memory_dim, mem.raw_msg_dim, mem.time_dim
        W = 2 * M + D + T
        # ---- TGNMemory.forward (train mode, reuse_forward): the look-ahead rows + the node -> row association --------------------------
        mem._ensure_store(2 * n)  # (a compaction, if one is due, runs now: the log's pointers are final below)
        if mem._assoc64 is None or mem._assoc64.device != dev:
            mem._assoc64 = torch.zeros(mem.num_nodes, dtype=torch.int64, device=dev)
            mem._reuse_status = torch.zeros(1, dtype=torch.int32, device=dev)
        stamp = md['_stamp'] = md['_stamp'] + 1
        ws = self._ws = self._grow(self._ws, R * (W + 7 * M), torch.float32, dev)
        z = torch.empty((R, M), dtype=torch.float32, device=dev)
        lu = torch.empty(R, dtype=torch.int64, device=dev)
        a = self._mem_args
        wkey = (param_key(mem), param_key(emb))
        if self._wkey != wkey:  # the weights' addresses (and the stacked projections) only move with the parameters
            gru = mem.memory_updater
            self._w = (mem.time_enc.w.weight.detach().reshape(-1), mem.time_enc.w.bias.detach(), gru.weight_ih.detach(), gru.bias_ih.detach(),
                       gru.weight_hh.detach(), gru.bias_hh.detach(), emb.time_enc.w.weight.detach().reshape(-1), emb.time_enc.w.bias.detach(),
                       conv.lin_edge.weight.detach()) + conv._stacked_projections()
            self._wkey = wkey
        tw, tb, W_ih, b_ih, W_hh, b_hh, etw, etb, W_edge, W4, b4 = self._w
        # ---- GraphAttentionEmbedding.forward: edge encoding + TransformerConv (inference) ---------------------------------------------
        U, H, C = R, conv.heads, conv.out_channels
        HC = H * C
        ei = i64c(edge_index) if edge_index.dtype != torch.int64 else edge_index
        E = ei.shape[1]
        src_e, tgt_e = ei[0], ei[1]
        if not src_e.is_contiguous():
            src_e = src_e.contiguous()
        if not tgt_e.is_contiguous():
            tgt_e = tgt_e.contiguous()
        et64, msg = i64c(edge_t), f32c(edge_x)
        De, Te = msg.shape[1], emb._modules['time_enc'].time_dim
        qkvs = torch.empty((4, U, HC), dtype=torch.float32, device=dev)
        lib = self._lib
        if lib is None:
            lib = self._lib = _native.load()
        wsd = getattr(conv, '_seg_ws', None)
        if wsd is None or E > self._seg_ws_edges or wsd[0].numel() < self._seg_ws_bytes or wsd[0].device != dev:
            # (the workspace bound is monotone in E: asked again only for a larger edge list than any before, with 25 % of slack)
            cap = max(E + E // 4, self._seg_ws_edges)
            need = int(lib.tgmx_segment_sort_workspace_bytes(cap))
            if wsd is None or wsd[0].device != dev or wsd[0].numel() < need:
                wsd = conv._seg_ws = (torch.empty(need, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
            self._seg_ws_edges, self._seg_ws_bytes = cap, need
        Ep, Up = E + (E & 1), U + (U & 1)
        fl = self._fl = self._grow(self._fl, E * (Te + De + HC), torch.float32, dev)
        ints = self._ints = self._grow(self._ints, 2 * Ep + 3 * Up, torch.int64, dev)
        cnt = getattr(conv, '_tgt_count', None)
        if cnt is None or cnt.device != dev or cnt.numel() < U:
            cnt = conv._tgt_count = torch.zeros(max(2 * U, 1 << 14), dtype=torch.int32, device=dev)
        c = self._conv_args
        s = self._args
        base, flp, ip = ws.data_ptr(), fl.data_ptr(), ints.data_ptr()
        # Everything that only moves when a buffer is (re)allocated or the parameters change is written into the argument blocks ONCE per
        # such event: ~60 ctypes field stores and ~25 data_ptr() calls less per batch (the pipeline is host-bound: every microsecond of
        # this function is a microsecond of the batch).  The key holds the address of every buffer a static field points into.
        aggr_mean = (mem._modules.get('aggr_module') or mem.aggr_module).mean
        skey = (wkey, bufs['memory'].data_ptr(), bufs['last_update'].data_ptr(), mem._st_lo[0].data_ptr(), mem._st_cnt[0].data_ptr(), mem._st_lo[1].data_ptr(),
                mem._st_cnt[1].data_ptr(), mem._log_other.data_ptr(), mem._log_t.data_ptr(), mem._log_raw.data_ptr(), mem._assoc64.data_ptr(),
                mem._reuse_status.data_ptr(), wsd[0].data_ptr(), wsd[0].numel(), wsd[1].data_ptr(), cnt.data_ptr(), M, D, T, De, Te, H, C,
                mem.num_nodes, aggr_mean)  # fmt: skip
        if skey != self._static_key:
            a.memory, a.last_update, a.M, a.num_nodes = skey[1], skey[2], M, mem.num_nodes
            a.st_lo_s, a.st_cnt_s, a.st_lo_d, a.st_cnt_d = skey[3], skey[4], skey[5], skey[6]
            a.log_other, a.log_t, a.log_raw, a.D = skey[7], skey[8], skey[9], D
            a.tw, a.tb, a.T, a.mean = tw.data_ptr(), tb.data_ptr(), T, aggr_mean
            a.W_ih, a.b_ih, a.W_hh, a.b_hh = W_ih.data_ptr(), b_ih.data_ptr(), W_hh.data_ptr(), b_hh.data_ptr()
            a.assoc = skey[10]
            c.in_ch, c.D, c.T = M, De, Te
            c.tw, c.tb, c.W4, c.b4, c.W_edge, c.H, c.C = etw.data_ptr(), etb.data_ptr(), W4.data_ptr(), b4.data_ptr(), W_edge.data_ptr(), H, C
            c.sort_ws, c.sort_ws_bytes, c.status = skey[12], skey[13], skey[14]
            c.tgt_count = skey[15]
            s.mem, s.conv = ctypes.addressof(a), ctypes.addressof(c)
            s.memory, s.last_update, s.reuse_status = a.memory, a.last_update, skey[11]
            s.log_other, s.log_t, s.log_raw = a.log_other, a.log_t, a.log_raw
            s.st_lo_s, s.st_cnt_s, s.st_lo_d, s.st_cnt_d = a.st_lo_s, a.st_cnt_s, a.st_lo_d, a.st_cnt_d
            self._static_key = skey
        # ---- what changes with every batch ---------------------------------------------------------------------------------------------
        a.nodes, a.R = nodes.data_ptr(), R
        a.ws_aggr, a.ws_h, a.ws_gi, a.ws_gh = base, base + 4 * R * W, base + 4 * R * (W + M), base + 4 * R * (W + 4 * M)
        a.out_mem, a.out_lu, a.stamp = z.data_ptr(), lu.data_ptr(), stamp
        c.x, c.U, c.last_update_local = a.out_mem, U, a.out_lu
        c.src, c.tgt, c.t, c.msg, c.E = src_e.data_ptr(), tgt_e.data_ptr(), et64.data_ptr(), msg.data_ptr(), E
        c.edge_attr, c.qkvs, c.eproj = flp, qkvs.data_ptr(), flp + 4 * E * (Te + De)
        c.order, c.seg_lo, c.seg_hi = ip, ip + 8 * Ep, ip + 8 * (Ep + Up)
        c.cursor, c.order_big = ip + 8 * (Ep + 2 * Up), ip + 8 * (Ep + 3 * Up)
        # ---- update_state (reuse_forward): commit the rows above for the batch's endpoints, store the batch ----------------------------
        s.src, s.dst, s.t, s.raw, s.n = src32.data_ptr(), dst32.data_ptr(), t64.data_ptr(), _native.ptr(raw), n
        s.log_base = mem._log_len
        _native.check(lib.tgmx_tgn_step(s, _native.stream_ptr()), 'tgmx_tgn_step')
        # what the three calls leave behind: two state mutations (commit, store), the log grown by both roles' entries, no pending forward
        md['_version'] += 2
        md['_log_len'] += 2 * n
        md['_fwd'] = None
        conv.__dict__['_edge_ctx'] = None
        return qkvs[3], z, lu
