"""CPU oracle of the GIN encoder + MoCo/InfoNCE head + train step -- TEST
INFRASTRUCTURE ONLY (plain torch on CPU; nothing under gcc_amd/ imports it).

Restates, with plain torch ops and the reference's state_dict keys:
  GraphEncoder.forward          gcc/models/graph_encoder.py:132-200
  UnsupervisedGIN / MLP / ApplyNodeFunc   gcc/models/gin.py:42-58,107-116,213-232
  DGL GINConv(sum, eps=0) / SumPooling    call sites gin.py:179-185,205,218,228
  MemoryMoCo.forward            gcc/contrastive/memory_moco.py:26-63
  NCESoftmaxLoss / NCESoftmaxLossNS       gcc/contrastive/criterions.py:5-33
  moment_update, clip, lr schedule        train.py:169-172,340-347,409-417; gcc/utils/misc.py:5-10

Pinned: tests/test_oracle_encoder.py checks this file against
tests/golden/encoder_golden.pt, which was produced by executing the
reference's own gin.py / graph_encoder.py / memory_moco.py / criterions.py
(DGL stubbed by tests/golden/dgl_stub.py -- the GINConv/SumPooling semantics
themselves are "DGL-recalled", i.e. unpinned).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _MLP(nn.Module):                      # gin.py:61-116 with num_layers == 2
    def __init__(self, d_in, d_hid, d_out):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(d_in, d_hid), nn.Linear(d_hid, d_out)])
        self.batch_norms = nn.ModuleList([nn.BatchNorm1d(d_hid)])


class _Apply(nn.Module):                    # gin.py:42-58
    def __init__(self, mlp, d):
        super().__init__()
        self.mlp = mlp
        self.bn = nn.BatchNorm1d(d)


class _GINLayer(nn.Module):                 # DGL GINConv container: apply_func + eps buffer
    def __init__(self, apply_func):
        super().__init__()
        self.apply_func = apply_func
        self.register_buffer("eps", torch.FloatTensor([0]))


class _GIN(nn.Module):                      # gin.py:119-211
    def __init__(self, num_layers, d_in, d_hid, d_out):
        super().__init__()
        self.ginlayers = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        for layer in range(num_layers - 1):
            self.ginlayers.append(_GINLayer(_Apply(_MLP(d_in if layer == 0 else d_hid, d_hid, d_hid), d_hid)))
            self.batch_norms.append(nn.BatchNorm1d(d_hid))
        self.linears_prediction = nn.ModuleList(
            [nn.Linear(d_in if layer == 0 else d_hid, d_out) for layer in range(num_layers)])


class _Set2Set(nn.Module):                  # DGL Set2Set: parameters only (unused by GIN, graph_encoder.py:189-194)
    def __init__(self, d, n_layers):
        super().__init__()
        self.lstm = nn.LSTM(2 * d, d, n_layers)


class OracleGraphEncoder(nn.Module):
    """gnn_model="gin", degree_input=True branch of graph_encoder.py:44-200."""

    def __init__(self, positional_embedding_size=32, max_degree=512, degree_embedding_size=16,
                 output_dim=64, node_hidden_dim=64, num_layers=5, num_layer_set2set=3, norm=True,
                 final_dropout=0.5):
        super().__init__()
        d_in = positional_embedding_size + degree_embedding_size + 1          # :66-67
        self.gnn = _GIN(num_layers, d_in, node_hidden_dim, output_dim)
        self.degree_embedding = nn.Embedding(max_degree + 1, degree_embedding_size)   # :116-118
        self.set2set = _Set2Set(node_hidden_dim, num_layer_set2set)            # :124
        self.lin_readout = nn.Sequential(nn.Linear(2 * node_hidden_dim, node_hidden_dim), nn.ReLU(),
                                         nn.Linear(node_hidden_dim, output_dim))   # :125-129
        self.max_degree = max_degree
        self.norm = norm
        self.p = final_dropout

    def forward(self, node_off, row_ptr, col_idx, pos_undirected, dropout_masks=None,
                return_all_outputs=False, seed_local=None):
        """dropout_masks: None (eval / no dropout) or float tensor [L, B, out] of keep masks (0/1)."""
        node_off = torch.as_tensor(node_off, dtype=torch.long)
        row_ptr = torch.as_tensor(row_ptr, dtype=torch.long)
        col_idx = torch.as_tensor(col_idx, dtype=torch.long)
        n, B = int(node_off[-1]), len(node_off) - 1
        src = torch.repeat_interleave(torch.arange(n), row_ptr[1:] - row_ptr[:-1])
        dst = col_idx
        gid = torch.repeat_interleave(torch.arange(B), node_off[1:] - node_off[:-1])
        seed = torch.zeros(n, dtype=pos_undirected.dtype)                       # data_util.py:234-238
        first = node_off[:-1] + (0 if seed_local is None else torch.as_tensor(seed_local, dtype=torch.long))
        seed[first[node_off[1:] > node_off[:-1]]] = 1.0                        # (empty padding graphs have no seed)
        degrees = torch.bincount(dst, minlength=n)                              # g.in_degrees(), :154
        h = torch.cat((pos_undirected, self.degree_embedding(degrees.clamp(0, self.max_degree)),
                       seed.unsqueeze(1)), dim=-1)                              # :158-165
        g = self.gnn
        hidden = [h]
        for i, layer in enumerate(g.ginlayers):                                 # gin.py:217-221
            neigh = torch.zeros_like(h).index_add_(0, dst, h[src])             # copy_u -> sum over in-edges
            x = (1 + layer.eps) * h + neigh
            mlp = layer.apply_func.mlp
            x = mlp.linears[1](F.relu(mlp.batch_norms[0](mlp.linears[0](x))))  # gin.py:113-116
            x = F.relu(layer.apply_func.bn(x))                                  # gin.py:55-57
            h = F.relu(g.batch_norms[i](x))                                     # gin.py:219-220
            hidden.append(h)
        score, all_outputs = 0, []
        for i, hh in enumerate(hidden):                                         # gin.py:227-230
            pooled = torch.zeros(B, hh.shape[1], dtype=hh.dtype).index_add_(0, gid, hh)
            all_outputs.append(pooled)
            y = g.linears_prediction[i](pooled)
            if dropout_masks is not None:
                y = y * dropout_masks[i] / (1.0 - self.p)
            score = score + y
        if self.norm:
            score = F.normalize(score, p=2, dim=-1, eps=1e-5)                   # graph_encoder.py:195-196
        return (score, all_outputs[1:]) if return_all_outputs else score


def moco_forward(memory, index, q, k, T):
    """memory_moco.py:26-63 with use_softmax=True -> (out[B, K+1], new_index); enqueues k in place."""
    B, K = q.shape[0], memory.shape[0]
    k = k.detach()
    l_pos = (q * k).sum(dim=1, keepdim=True)
    l_neg = q @ memory.clone().detach().t()
    out = torch.cat((l_pos, l_neg), dim=1) / T
    with torch.no_grad():
        ids = (torch.arange(B) + index) % K
        memory.index_copy_(0, ids, k)
    return out, (index + B) % K


def nce_softmax_loss(out):
    """criterions.py:12-17 (label 0)."""
    return F.cross_entropy(out, torch.zeros(out.shape[0], dtype=torch.long))


def nce_softmax_loss_ns(out):
    """criterions.py:27-33 (labels on the diagonal)."""
    return F.cross_entropy(out, torch.arange(out.shape[0]))


def moment_update(model, model_ema, m):
    """train.py:169-172."""
    for p1, p2 in zip(model.parameters(), model_ema.parameters()):
        p2.data.mul_(m).add_(p1.detach().data, alpha=1 - m)


def warmup_linear(x, warmup=0.002):
    """gcc/utils/misc.py:5-10."""
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0)


def memory_init(K, d, generator=None):
    """memory_moco.py:20-23."""
    stdv = 1.0 / math.sqrt(d / 3)
    return torch.rand(K, d, generator=generator).mul_(2 * stdv).add_(-stdv)
