"""A minimal stand-in for the parts of DGL 0.4.x that /root/reference imports,
so that the reference's OWN model code (gcc/models/gin.py, graph_encoder.py,
gcc/contrastive/*.py) can be executed in this container to generate golden
vectors (tests/golden/make_encoder_golden.py).  DGL is an un-vendored pip
dependency (README.md:45, 0.5 > dgl >= 0.4.3) and is not installed here.

What is restated from the published dgl 0.4.3 source (NOT pinned by anything in
/root/reference -- "DGL-recalled" in SURVEY.md):
  GINConv.forward : rst = (1 + eps) * feat + sum_{u -> v} feat[u]; apply_func(rst)
                    eps is a registered buffer when learn_eps=False
  SumPooling      : per-graph sum of node features of a batched graph
  Set2Set         : only its parameters (an LSTM(2d, d, n_layers)) matter here;
                    GraphEncoder allocates it but the GIN path never calls it
Everything else the reference files touch at import time is a placeholder.
"""
import sys
import types

import torch
import torch.nn as nn


class StubBatchedGraph:
    """The handful of DGLGraph members the hot path uses."""

    def __init__(self, node_off, row_ptr, col_idx, pos_undirected):
        self.node_off = torch.as_tensor(node_off, dtype=torch.long)
        self.row_ptr = torch.as_tensor(row_ptr, dtype=torch.long)
        self.col_idx = torch.as_tensor(col_idx, dtype=torch.long)
        n = int(self.node_off[-1])
        self.batch_size = len(self.node_off) - 1
        seed = torch.zeros(n, dtype=torch.long)
        seed[self.node_off[:-1]] = 1                       # data_util.py:234-238
        self.ndata = {"pos_undirected": torch.as_tensor(pos_undirected, dtype=torch.float32), "seed": seed}
        # edge list u -> v in CSR order: row u lists its successors v
        self.src = torch.repeat_interleave(torch.arange(n), self.row_ptr[1:] - self.row_ptr[:-1])
        self.dst = self.col_idx
        self.graph_id = torch.repeat_interleave(torch.arange(self.batch_size), self.node_off[1:] - self.node_off[:-1])

    def number_of_nodes(self):
        return int(self.node_off[-1])

    def number_of_edges(self):
        return int(self.col_idx.numel())

    def in_degrees(self):
        return torch.bincount(self.dst, minlength=self.number_of_nodes())

    def to(self, device):
        return self


class GINConv(nn.Module):
    def __init__(self, apply_func, aggregator_type, init_eps=0, learn_eps=False):
        super().__init__()
        assert aggregator_type == "sum"
        self.apply_func = apply_func
        if learn_eps:
            self.eps = nn.Parameter(torch.FloatTensor([init_eps]))
        else:
            self.register_buffer("eps", torch.FloatTensor([init_eps]))

    def forward(self, graph, feat):
        neigh = torch.zeros_like(feat).index_add_(0, graph.dst, feat[graph.src])   # update_all(copy_u, sum)
        rst = (1 + self.eps) * feat + neigh
        if self.apply_func is not None:
            rst = self.apply_func(rst)
        return rst


class SumPooling(nn.Module):
    def forward(self, graph, feat):
        out = torch.zeros(graph.batch_size, feat.shape[1], dtype=feat.dtype)
        return out.index_add_(0, graph.graph_id, feat)


class AvgPooling(SumPooling):
    pass


class MaxPooling(SumPooling):
    pass


class Set2Set(nn.Module):
    def __init__(self, input_dim, n_iters, n_layers):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, 2 * input_dim
        self.n_iters, self.n_layers = n_iters, n_layers
        self.lstm = nn.LSTM(self.output_dim, self.input_dim, n_layers)


class _Placeholder(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def install():
    """Register fake ``dgl`` modules in sys.modules (idempotent)."""
    if "dgl" in sys.modules and getattr(sys.modules["dgl"], "_gcc_amd_stub", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    dgl = mod("dgl", _gcc_amd_stub=True, DGLGraph=StubBatchedGraph, batch=None)
    mod("dgl.function", copy_u=None, sum=None)
    dgl.function = sys.modules["dgl.function"]
    mod("dgl.nn")
    mod("dgl.nn.pytorch", Set2Set=Set2Set, NNConv=_Placeholder)
    mod("dgl.nn.pytorch.conv", GINConv=GINConv)
    mod("dgl.nn.pytorch.glob", AvgPooling=AvgPooling, MaxPooling=MaxPooling, SumPooling=SumPooling)
    mod("dgl.model_zoo")
    mod("dgl.model_zoo.chem")
    mod("dgl.model_zoo.chem.gnn", GATLayer=_Placeholder)
    mod("dgl.data", AmazonCoBuy=None, Coauthor=None)
    mod("dgl.data.tu", TUDataset=None)
    mod("dgl.nodeflow", NodeFlow=None)
    dgl.nn = sys.modules["dgl.nn"]
    dgl.nn.pytorch = sys.modules["dgl.nn.pytorch"]
    dgl.data = sys.modules["dgl.data"]
