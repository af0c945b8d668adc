"""ORACLE (test infrastructure only -- never imported by the product path).

Plain torch-fp32 restatement of the in-tree part of the reference's TGN memory
(/root/reference/tgm/nn/encoder/tgn.py):
  :43-56    LastAggregator  -- per node the message with the largest float32(t), first max wins
  :59-63    MeanAggregator  -- mean of the node's messages
  :66-74    IdentityMessage -- [mem[self] | mem[other] | raw_msg | Time2Vec(t - last_update[self])]
  :157-163  forward         -- train: look-ahead memory (nothing written); eval: table read
  :165-177  update_state    -- train: commit the STORED messages, then replace the store with this batch;
                               eval: replace the store first, then commit
  :191-216  _get_updated_memory -- GRUCell(aggr, memory[n_id]) for EVERY n_id (aggr = 0 without messages),
                               last_update = max stored t (0 without messages)
  :218-229  _update_msg_store -- a node's store is REPLACED by its events of this batch, per role
  :245-251  train(False)    -- commit all N nodes, clear the stores

A node's stored events are kept here in batch (edge) order; the reference orders them with a
non-stable sort (tgn.py:226), i.e. unspecified when one node has several events in a batch --
only observable when two of them share the same float32 timestamp (LastAggregator tie).

Parity status: pinned against tests/golden/g8_tgn_*.npz (recorded from the reference with the
placeholder `scatter`) by tests/test_tgn_oracle_cpu.py.  The PyG TransformerConv behind
GraphAttentionEmbedding (tgn.py:25-27) is third-party arithmetic: parity unpinned.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
from torch import Tensor

from .tgat_ref import time2vec


class TGNMemoryRef:
    def __init__(self, num_nodes: int, raw_msg_dim: int, memory_dim: int, time_dim: int, params: Dict[str, Tensor], aggr: str = 'last'):
        self.N, self.D, self.M, self.T = num_nodes, raw_msg_dim, memory_dim, time_dim
        self.p = params
        self.aggr = aggr
        self.training = True
        self.memory = torch.zeros(num_nodes, memory_dim)
        self.last_update = torch.zeros(num_nodes, dtype=torch.int64)
        self._clear_store()

    def _clear_store(self) -> None:
        self.store: List[Dict[int, Tuple[Tensor, Tensor, Tensor]]] = [dict(), dict()]  # role -> node -> (other, t, raw)

    def reset_state(self) -> None:
        self.memory.zero_()
        self.last_update.zero_()
        self._clear_store()

    # ------------------------------------------------------------------
    def _gru(self, x: Tensor, h: Tensor) -> Tensor:
        p, M = self.p, self.M
        gi = x @ p['memory_updater.weight_ih'].T + p['memory_updater.bias_ih']
        gh = h @ p['memory_updater.weight_hh'].T + p['memory_updater.bias_hh']
        r = torch.sigmoid(gi[:, :M] + gh[:, :M])
        z = torch.sigmoid(gi[:, M : 2 * M] + gh[:, M : 2 * M])
        n = torch.tanh(gi[:, 2 * M :] + r * gh[:, 2 * M :])
        return (1 - z) * n + z * h

    def _updated(self, n_id: Tensor) -> Tuple[Tensor, Tensor]:
        msg_dim = 2 * self.M + self.D + self.T
        aggr = torch.zeros(len(n_id), msg_dim)
        new_lu = torch.zeros(len(n_id), dtype=torch.int64)
        for row, v in enumerate(n_id.tolist()):
            msgs, times = [], []
            for role in (0, 1):  # source-role store first, then destination-role (tgn.py:196-209)
                if v in self.store[role]:
                    other, t, raw = self.store[role][v]
                    t_enc = time2vec(t - self.last_update[v], self.p['time_enc.w.weight'], self.p['time_enc.w.bias'])
                    msgs.append(torch.cat([self.memory[v].expand(len(t), -1), self.memory[other.long()], raw, t_enc], dim=1))
                    times.append(t)
            if msgs:
                m, t = torch.cat(msgs), torch.cat(times)
                new_lu[row] = t.max()
                aggr[row] = m[int(torch.argmax(t.float()))] if self.aggr == 'last' else m.mean(0)
        return self._gru(aggr, self.memory[n_id]), new_lu

    def forward(self, n_id: Tensor) -> Tuple[Tensor, Tensor]:
        if self.training:
            return self._updated(n_id)
        return self.memory[n_id], self.last_update[n_id]

    def _commit(self, n_id: Tensor) -> None:
        mem, lu = self._updated(n_id)
        self.memory[n_id] = mem
        self.last_update[n_id] = lu

    def _store(self, role: int, node: Tensor, other: Tensor, t: Tensor, raw: Tensor) -> None:
        for v in torch.unique(node).tolist():
            sel = node == v  # batch order
            self.store[role][v] = (other[sel], t[sel], raw[sel])

    def update_state(self, src: Tensor, dst: Tensor, t: Tensor, raw: Tensor) -> None:
        n_id = torch.unique(torch.cat([src, dst])).long()
        if self.training:
            self._commit(n_id)
            self._store(0, src, dst, t, raw)
            self._store(1, dst, src, t, raw)
        else:
            self._store(0, src, dst, t, raw)
            self._store(1, dst, src, t, raw)
            self._commit(n_id)

    def eval(self) -> None:
        if self.training:
            self._commit(torch.arange(self.N))
            self._clear_store()
        self.training = False

    def train(self) -> None:
        self.training = True


def transformer_conv_ref(p: Dict[str, Tensor], prefix: str, heads: int, x: Tensor, edge_index: Tensor, edge_attr: Tensor, dropout=None) -> Tensor:
    """torch_geometric.nn.TransformerConv (2.6.1) from its published definition -- concat heads,
    root_weight=True, beta=False, eval mode:  out_i = W_skip x_i + b + ||_h sum_j alpha^h_ij (W_v x_j + b_v + W_e e_ij),
    alpha^h_ij = softmax over the edges j->i of (W_q x_i + b_q)^h . (W_k x_j + b_k + W_e e_ij)^h / sqrt(C).
    UNPINNED: PyG is not installable here, the reference's own tests only check shapes.
    ``dropout`` = (p, seed, stream): train mode -- PyG applies F.dropout to alpha after the softmax; the mask is the
    kernels' counter-based one, element (edge e, head h) -> e * heads + h (oracle/dropout_ref.py)."""
    lin = lambda name, v: v @ p[prefix + name + '.weight'].T + (p[prefix + name + '.bias'] if prefix + name + '.bias' in p else 0)
    HC = p[prefix + 'lin_query.weight'].shape[0]
    C = HC // heads
    q, k, v = lin('lin_query', x), lin('lin_key', x), lin('lin_value', x)
    e = lin('lin_edge', edge_attr)
    src, tgt = edge_index[0].long(), edge_index[1].long()
    key = (k[src] + e).view(-1, heads, C)
    val = (v[src] + e).view(-1, heads, C)
    score = (q[tgt].view(-1, heads, C) * key).sum(-1) / C**0.5  # [E, H]
    out = lin('lin_skip', x).clone()
    keep = None
    if dropout is not None and dropout[0]:
        from .dropout_ref import dropout_scale

        keep = dropout_scale(dropout[0], dropout[1], dropout[2], (score.shape[0], heads)).to(score.dtype)
    for i in torch.unique(tgt).tolist():
        m = tgt == i
        a = torch.softmax(score[m], dim=0)  # over the incoming edges of i
        if keep is not None:
            a = a * keep[m]
        out[i] += (a[:, :, None] * val[m]).sum(0).reshape(-1)
    return out


def graph_attention_embedding_ref(p: Dict[str, Tensor], x, last_update, edge_index, t, msg, dropout=None) -> Tensor:
    rel_t = last_update[edge_index[0].long()] - t
    enc = time2vec(rel_t, p['time_enc.w.weight'], p['time_enc.w.bias'])
    return transformer_conv_ref(p, 'conv.', 2, x, edge_index, torch.cat([enc, msg], dim=-1), dropout=dropout)
