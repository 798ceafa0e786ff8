"""Parent graph resident in HBM.

The reference keeps one DGLGraph copy per DataLoader worker process
(/root/reference/gcc/datasets/graph_dataset.py:23-30); here the CSR is uploaded
once per GPU and every kernel reads it in place.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _cabi
from .graphgen import check_contract


def seed_cdf_table(row_ptr: np.ndarray, shard_off=None) -> np.ndarray:
    """P(seed = v) ~ in_degree(v)^0.75 (graph_dataset.py:86-90) as the float64
    cdf numpy's ``choice(p=...)`` searches (``cdf = p.cumsum(); cdf /= cdf[-1]``).
    ``shard_off``: node ranges of the DataLoader workers' shards (graph_dataset.py:23-30,63-76) --
    a worker normalises the weights over ITS graphs, so every range gets its own cdf."""
    w = np.diff(row_ptr).astype(np.float64) ** 0.75
    bounds = [0, len(w)] if shard_off is None or len(shard_off) <= 2 else [int(x) for x in shard_off]
    cdf = np.empty(len(w), dtype=np.float64)
    for a, b in zip(bounds[:-1], bounds[1:]):
        tot = w[a:b].sum()
        if not (np.isfinite(tot) and tot > 0):          # 0/0 would give a NaN cdf and every seed the shard's last node
            raise ValueError(f"seed cdf: the nodes [{a}, {b}) of a worker shard have no edges (total deg^0.75 weight {tot})")
        p = w[a:b] / tot
        c = np.cumsum(p)
        cdf[a:b] = c / c[-1]
    return cdf


def max_nodes_per_seed_table(max_degree: int, rw_hops: int, restart_prob: float) -> np.ndarray:
    """max_nodes_per_seed of graph_dataset.py:113-124 for every in-degree 0..max_degree."""
    # (the reference's order of operations, so that the rounding at x.5 cannot differ: pinned by the reference-executed
    #  tests/golden/getitem_calls_reference.json)
    return np.array([max(rw_hops, int((d ** 0.75) * math.e / (math.e - 1) / restart_prob + 0.5)) for d in range(max_degree + 1)],
                    dtype=np.int32)


def max_nodes_out_degree_table(max_degree: int, rw_hops: int, restart_prob: float, multiplicity: int = 1) -> np.ndarray:
    """max_nodes_per_seed of the GraphDataset family (graph_dataset.py:244-255): the out-degree enters WITHOUT the
    0.75 power; ``multiplicity`` = copies of every edge in the DGL graph the reference walks on."""
    # (the reference's order of operations; pinned by tests/golden/getitem_calls_reference.json)
    return np.array([max(rw_hops, int(d * multiplicity * math.e / (math.e - 1) / restart_prob + 0.5)) for d in range(max_degree + 1)],
                    dtype=np.int32)


HUB_TABLE_DEGREE = 512      # = the sampler's default hub_degree (csrc/sampler.hip kHubDegreeDefault)


def hub_tables(row_ptr: np.ndarray, col_idx: np.ndarray, degree: int = HUB_TABLE_DEGREE, max_bytes: int = 1 << 30):
    """Adjacency among the rows of at least ``degree`` entries (gcc_graph.hub_index / hub_adj): -> (hub_index int32[V],
    hub_adj uint32[H, words]) or None when there are no such rows or the bitmap would exceed ``max_bytes``.  One pass over
    the hub rows' entries (1.6 M on the 1M-node graph, 49 M on the 10M-node one: 1286 / 29,784 hubs, 0.2 / 111 MB)."""
    deg = np.diff(row_ptr)
    hubs = np.flatnonzero(deg >= degree)
    H = int(len(hubs))
    words = (H + 31) // 32
    if H == 0 or H * words * 4 > max_bytes:
        return None
    index = np.full(len(deg), -1, dtype=np.int32)
    index[hubs] = np.arange(H, dtype=np.int32)
    adj = np.zeros((H, words), dtype=np.uint32)
    # entries of the hub rows, row by row in slabs (bounded temporaries)
    starts, lens = row_ptr[hubs].astype(np.int64), deg[hubs].astype(np.int64)
    at = 0
    while at < H:
        end = at
        tot = 0
        while end < H and (tot == 0 or tot + lens[end] <= (1 << 24)):
            tot += int(lens[end])
            end += 1
        rows = np.repeat(np.arange(at, end, dtype=np.int64), lens[at:end])
        offs = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(lens[at:end]) - lens[at:end], lens[at:end])
        cols = index[col_idx[np.repeat(starts[at:end], lens[at:end]) + offs]]
        keep = cols >= 0
        r, c = rows[keep], cols[keep].astype(np.int64)
        np.bitwise_or.at(adj, (r, c >> 5), (np.uint32(1) << (c & 31).astype(np.uint32)))
        at = end
    return index, adj


def restart_threshold(restart_prob: float) -> int:
    """restart <=> 32-bit draw < floor(restart_prob * 2^32)."""
    return min(int(restart_prob * 4294967296.0), 0xFFFFFFFF)


class DeviceGraph:
    """int32 CSR + seed cdf + max_nodes table on one GPU."""

    def __init__(self, row_ptr: np.ndarray, col_idx: np.ndarray, rw_hops: int = 256,
                 restart_prob: float = 0.8, device="cuda", validate: bool = True, ltab: np.ndarray = None,
                 shard_off=None, trusted: bool = False, hub_table: bool = True):
        """``validate``: check the input contract (x2dgl.py:39-62) on the host before the upload.  ``trusted``: the caller
        vouches for the contract without the check (graphs of gcc_amd.graphgen, whose generator builds to it).  With
        neither, the graph carries no GCC_GRAPH_CONTRACT_CHECKED bit and the induction scans every member row (the hub-row
        short cut of gcc_sample_params.hub_degree is exact on a symmetric, sorted, duplicate- and loop-free parent only)."""
        import torch

        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        if validate:
            check_contract(row_ptr, col_idx)
        self.contract_checked = bool(validate or trusted)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceGraph lives in HBM; device must be a HIP/CUDA device")
        self.num_nodes = int(row_ptr.shape[0] - 1)
        self.num_edges = int(col_idx.shape[0])
        self.rw_hops = int(rw_hops)
        self.restart_prob = float(restart_prob)
        self.restart_u32 = restart_threshold(restart_prob)
        deg = np.diff(row_ptr)
        self.max_degree = int(deg.max())
        d64 = deg.astype(np.float64)
        self.sb_degree = float((d64 * d64).sum() / d64.sum())   # size-biased mean degree: what a random-walk visit sees
        if ltab is None:
            ltab = max_nodes_per_seed_table(self.max_degree, rw_hops, restart_prob)
        ltab = np.ascontiguousarray(ltab, dtype=np.int32)
        assert ltab.shape[0] == self.max_degree + 1
        self.lmax = int(ltab.max())
        self.row_ptr = torch.from_numpy(row_ptr).to(self.device)
        self.col_idx = torch.from_numpy(col_idx).to(self.device)
        # worker shards (LoadBalanceGraphDataset): node ranges whose seeds are drawn separately, batch by batch
        self.num_shards = 0
        self.shard_off = None
        if shard_off is not None and len(shard_off) > 2:
            so = np.ascontiguousarray(shard_off, dtype=np.int64)
            if so[0] != 0 or so[-1] != self.num_nodes or np.any(np.diff(so) <= 0):
                raise ValueError("shard_off must be increasing node offsets from 0 to num_nodes (no empty shard)")
            self.num_shards = len(so) - 1
            self.shard_off = torch.from_numpy(so).to(self.device)
        self.seed_cdf = torch.from_numpy(seed_cdf_table(row_ptr, shard_off if self.num_shards else None)).to(self.device)
        self.ltab = torch.from_numpy(ltab).to(self.device)
        # hub-hub adjacency (one bit probe per pair of unscanned hub rows instead of a search): only with a checked contract
        # -- the hub-row short cut is off without it anyway
        self.hub_index = self.hub_adj = None
        self.num_hubs = self.hub_words = 0
        tabs = hub_tables(row_ptr, col_idx) if (self.contract_checked and hub_table) else None
        if tabs is not None:
            self.hub_index = torch.from_numpy(tabs[0]).to(self.device)
            self.hub_adj = torch.from_numpy(tabs[1].view(np.int32)).to(self.device)
            self.num_hubs, self.hub_words = int(tabs[1].shape[0]), int(tabs[1].shape[1])
        self.c = _cabi.GccGraph(
            hub_index=self.hub_index.data_ptr() if self.hub_index is not None else None,
            hub_adj=self.hub_adj.data_ptr() if self.hub_adj is not None else None,
            num_hubs=self.num_hubs, hub_words=self.hub_words, hub_table_degree=HUB_TABLE_DEGREE if self.num_hubs else 0,
            row_ptr=self.row_ptr.data_ptr(), col_idx=self.col_idx.data_ptr(),
            seed_cdf=self.seed_cdf.data_ptr(), ltab=self.ltab.data_ptr(),
            num_nodes=self.num_nodes, num_edges=self.num_edges,
            ltab_len=int(ltab.shape[0]), lmax=self.lmax,
            shard_off=self.shard_off.data_ptr() if self.shard_off is not None else None, num_shards=self.num_shards,
            flags=_cabi.GRAPH_CONTRACT_CHECKED if self.contract_checked else 0)

    def byref(self):
        return ctypes.byref(self.c)

    def hbm_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.row_ptr, self.col_idx, self.seed_cdf, self.ltab, self.hub_index, self.hub_adj)
                   if t is not None)
