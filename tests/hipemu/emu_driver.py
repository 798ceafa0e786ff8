"""Drives tests/hipemu/_build/libgcc_amd_emu.so (the kernels of gcc_amd/csrc
compiled for the lock-step CPU emulator) through the C ABI with numpy buffers.

TEST INFRASTRUCTURE ONLY -- lets the "not gpu" suite execute the kernel logic
on machines without a GPU.  gcc_amd itself never imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from gcc_amd import _cabi
from gcc_amd.graph import max_nodes_per_seed_table, restart_threshold, seed_cdf_table

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = os.path.join(_HERE, "_build", "libgcc_amd_emu.so")
_lib = None


def emu_lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", os.path.join(_ROOT, "gcc_amd", "csrc"), "emu"], check=True,
                       capture_output=True)
        _lib = _cabi.declare(ctypes.CDLL(_LIB))
    return _lib


def _p(a):
    return a.ctypes.data if a is not None else None


class EmuGraph:
    def __init__(self, row_ptr, col_idx, rw_hops=256, restart_prob=0.8, ltab=None, shard_off=None, contract_checked=True,
                 hub_table_degree=None):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.shard_off = np.ascontiguousarray(shard_off, dtype=np.int64) if shard_off is not None else None
        self.cdf = seed_cdf_table(self.row_ptr, self.shard_off)
        deg = np.diff(self.row_ptr)
        self.ltab = (max_nodes_per_seed_table(int(deg.max()), rw_hops, restart_prob) if ltab is None
                     else np.ascontiguousarray(ltab, dtype=np.int32))
        self.lmax = int(self.ltab.max())
        self.rw_hops = rw_hops
        self.restart_u32 = restart_threshold(restart_prob)
        self.c = _cabi.GccGraph(row_ptr=_p(self.row_ptr), col_idx=_p(self.col_idx), seed_cdf=_p(self.cdf),
                                ltab=_p(self.ltab), num_nodes=len(self.row_ptr) - 1,
                                num_edges=len(self.col_idx), ltab_len=len(self.ltab), lmax=self.lmax,
                                shard_off=_p(self.shard_off), num_shards=len(self.shard_off) - 1 if shard_off is not None else 0,
                                flags=_cabi.GRAPH_CONTRACT_CHECKED if contract_checked else 0)
        if hub_table_degree is not None:       # the hub-hub adjacency table (gcc_amd.graph.hub_tables) for this threshold
            from gcc_amd.graph import hub_tables

            tabs = hub_tables(self.row_ptr, self.col_idx, hub_table_degree)
            if tabs is not None:
                self.hub_index, self.hub_adj = tabs
                self.c.hub_index, self.c.hub_adj = _p(self.hub_index), _p(self.hub_adj)
                self.c.num_hubs, self.c.hub_words, self.c.hub_table_degree = self.hub_adj.shape[0], self.hub_adj.shape[1], hub_table_degree


def emu_sample_batch(g: EmuGraph, B, run_seed, first_sample_id, seeds=None, edge_cap=None,
                     scratch_entries=None, node_cap=None, hub_degree=0, max_hubs=0):
    lib = emu_lib()
    node_cap = node_cap or B * (g.lmax + 1)
    edge_cap = edge_cap or B * (g.lmax + 1) ** 2
    # induction scratch: 4 slots per aligned quad of every member row (include/gcc_amd.h)
    scratch_entries = scratch_entries or 2 * B * (g.lmax + 1) * (int(np.diff(g.row_ptr).max()) + 8)
    nbytes = lib.gcc_sampler_workspace_bytes(ctypes.byref(g.c), B, scratch_entries)
    assert nbytes > 0
    ws = np.zeros(nbytes, dtype=np.uint8)
    status = np.zeros(1, dtype=np.int32)
    outs, structs = [], []
    for _ in range(2):
        o = dict(node_off=np.zeros(B + 1, np.int32), edge_off=np.zeros(B + 1, np.int32),
                 parent_nid=np.zeros(node_cap, np.int32), graph_id=np.zeros(node_cap, np.int32),
                 row_ptr=np.zeros(node_cap + 1, np.int32), col_idx=np.zeros(edge_cap, np.int32))
        outs.append(o)
        structs.append(_cabi.GccBatchOut(node_off=_p(o["node_off"]), edge_off=_p(o["edge_off"]),
                                         parent_nid=_p(o["parent_nid"]), graph_id=_p(o["graph_id"]),
                                         row_ptr=_p(o["row_ptr"]), col_idx=_p(o["col_idx"]),
                                         node_cap=node_cap, edge_cap=edge_cap))
    if seeds is not None:
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
    params = _cabi.GccSampleParams(run_seed=run_seed, first_sample_id=first_sample_id, batch_size=B,
                                   restart_u32=g.restart_u32, seeds=_p(seeds), hub_degree=hub_degree, max_hubs=max_hubs)
    rc = lib.gcc_sample_batch(ctypes.byref(g.c), ctypes.byref(params), ctypes.byref(structs[0]),
                              ctypes.byref(structs[1]), _p(ws), nbytes, scratch_entries, _p(status), None)
    if rc != 0:
        raise RuntimeError(lib.gcc_last_error().decode())
    res = []
    for o in outs:
        n, e = int(o["node_off"][B]), int(o["edge_off"][B])
        res.append(dict(node_off=o["node_off"], edge_off=o["edge_off"], parent_nid=o["parent_nid"][:n],
                        graph_id=o["graph_id"][:n], row_ptr=o["row_ptr"][: n + 1], col_idx=o["col_idx"][:e]))
    return res, int(status[0]), ws[: 4 * B].view(np.int32).copy()


def emu_sample_multi(g: EmuGraph, B, run_seed, first_sample_id, num_steps, stride, edge_cap=None, scratch_entries=None,
                     node_cap=None, hub_degree=0, max_hubs=0):
    """gcc_sample_multi on the emulator -> ([(q, k) per step], status, seeds [num_steps * B])."""
    lib = emu_lib()
    node_cap = node_cap or B * (g.lmax + 1)
    edge_cap = edge_cap or B * (g.lmax + 1) ** 2
    scratch_entries = scratch_entries or 2 * num_steps * B * (g.lmax + 1) * (int(np.diff(g.row_ptr).max()) + 8)
    nbytes = lib.gcc_sampler_workspace_bytes_multi(ctypes.byref(g.c), B, num_steps, scratch_entries)
    assert nbytes > 0, lib.gcc_last_error().decode()
    ws = np.zeros(nbytes, dtype=np.uint8)
    status = np.zeros(1, dtype=np.int32)
    outs = []
    structs = (_cabi.GccBatchOut * (2 * num_steps))()
    for i in range(2 * num_steps):
        o = dict(node_off=np.zeros(B + 1, np.int32), edge_off=np.zeros(B + 1, np.int32),
                 parent_nid=np.zeros(node_cap, np.int32), graph_id=np.zeros(node_cap, np.int32),
                 row_ptr=np.zeros(node_cap + 1, np.int32), col_idx=np.zeros(edge_cap, np.int32))
        outs.append(o)
        structs[i] = _cabi.GccBatchOut(node_off=_p(o["node_off"]), edge_off=_p(o["edge_off"]),
                                       parent_nid=_p(o["parent_nid"]), graph_id=_p(o["graph_id"]),
                                       row_ptr=_p(o["row_ptr"]), col_idx=_p(o["col_idx"]),
                                       node_cap=node_cap, edge_cap=edge_cap)
    params = _cabi.GccSampleParams(run_seed=run_seed, first_sample_id=first_sample_id, batch_size=B,
                                   restart_u32=g.restart_u32, seeds=None, hub_degree=hub_degree, max_hubs=max_hubs)
    rc = lib.gcc_sample_multi(ctypes.byref(g.c), ctypes.byref(params), num_steps, stride, structs, _p(ws), nbytes,
                              scratch_entries, _p(status), None)
    if rc != 0:
        raise RuntimeError(lib.gcc_last_error().decode())
    res = []
    for o in outs:
        n, e = int(o["node_off"][B]), int(o["edge_off"][B])
        res.append(dict(node_off=o["node_off"], edge_off=o["edge_off"], parent_nid=o["parent_nid"][:n],
                        graph_id=o["graph_id"][:n], row_ptr=o["row_ptr"][: n + 1], col_idx=o["col_idx"][:e]))
    pairs = [(res[2 * t], res[2 * t + 1]) for t in range(num_steps)]
    return pairs, int(status[0]), ws[: 4 * B * num_steps].view(np.int32).copy()


def emu_ginw_forward(node_off, row_ptr, col_idx, x_bits, layers, pack=False, scratch=False):
    """gcc_ginw_forward on the emulator.  x_bits: uint16 [N, 256] bf16 patterns; layers: dicts of numpy arrays with
    w0/w1 as uint16 bf16 patterns [256, 256] and s0..t2 float32 [256].  ``pack``: also pass the fragment-major copies
    made by gcc_ginw_pack_weights.  ``scratch``: pass the scratch that lets subgraphs over 128 nodes run block by block.
    Returns (rows uint16, pooled f32, status)."""
    lib = emu_lib()
    node_off = np.ascontiguousarray(node_off, dtype=np.int32)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    x_bits = np.ascontiguousarray(x_bits, dtype=np.uint16)
    B, L = len(node_off) - 1, len(layers)
    rows = np.full_like(x_bits, 0xFFFF)
    pooled = np.full((B, L + 1, 256), np.nan, dtype=np.float32)
    status = np.zeros(1, dtype=np.int32)
    a = _cabi.GccGinwArgs(node_off=_p(node_off), row_ptr=_p(row_ptr), col_idx=_p(col_idx), x_in=_p(x_bits),
                          x_out=_p(rows), pooled=_p(pooled), batch_size=B, num_layers=L)
    keep = []
    for i, ly in enumerate(layers):
        for which, k in enumerate(("w0", "w1")):
            v = np.ascontiguousarray(ly[k], dtype=np.uint16)
            keep.append(v)
            setattr(a.layers[i], k, _p(v))
            if pack:
                f = np.zeros_like(v)
                assert lib.gcc_ginw_pack_weights(_p(v), _p(f), which, None) == 0
                keep.append(f)
                setattr(a.layers[i], k + "_frag", _p(f))
        for k in ("s0", "t0", "s1", "t1", "s2", "t2"):
            v = np.ascontiguousarray(ly[k], dtype=np.float32)
            keep.append(v)
            setattr(a.layers[i], k, _p(v))
    if scratch:
        nbytes = lib.gcc_ginw_scratch_bytes(len(x_bits), B)
        assert nbytes > 0
        sbuf = np.zeros(nbytes // 16 + 1, dtype=np.dtype([("a", np.uint64), ("b", np.uint64)]))      # 16-byte aligned
        keep.append(sbuf)
        a.scratch, a.scratch_bytes, a.num_nodes = _p(sbuf), nbytes, len(x_bits)
    rc = lib.gcc_ginw_forward(ctypes.byref(a), _p(status), None, None)
    if rc != 0:
        raise RuntimeError(lib.gcc_last_error().decode())
    return rows, pooled, int(status[0])
