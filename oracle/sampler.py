"""CPU oracle for the RWR ego-net sampler -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under gcc_amd/ does.  "Parity unpinned" at the DGL
boundary -- see the header of oracle/sampler_oracle.c for what is and is not
pinned and for the RNG spec.

Two independent restatements live here:
  * ``COracle`` -- ctypes binding of oracle/sampler_oracle.c (fast; used at
    bench sizes and as the CPU baseline),
  * ``py_*``   -- pure-Python loops over Python ints (slow; tiny cases only),
    written separately so that one can check the other.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

MASK32 = 0xFFFFFFFF


# --------------------------------------------------------------------------
# host-side tables (restated from the reference; compared with gcc_amd.graph)
# --------------------------------------------------------------------------
def seed_cdf(row_ptr: np.ndarray, shard_off=None) -> np.ndarray:
    """graph_dataset.py:86-90: p ~ in_degree^0.75 (float64), numpy choice() cdf.  With worker shards
    (graph_dataset.py:23-30,63-76: worker w samples among the nodes of ITS graphs only) every node range
    [shard_off[s], shard_off[s+1]) carries that shard's own cdf."""
    deg = np.diff(row_ptr).astype(np.float64) ** 0.75
    def _total(w, a, b):
        tot = w.sum()
        if not (np.isfinite(tot) and tot > 0):      # np.random.choice(p=NaN) raises in the reference; never a silent NaN cdf
            raise ValueError(f"seed cdf: the nodes [{a}, {b}) of a worker shard have no edges (total deg^0.75 weight {tot})")
        return tot

    if shard_off is None or len(shard_off) <= 2:
        p = deg / _total(deg, 0, len(deg))
        cdf = p.cumsum()
        cdf /= cdf[-1]
        return cdf
    out = np.empty(len(deg), dtype=np.float64)
    for s in range(len(shard_off) - 1):
        a, b = int(shard_off[s]), int(shard_off[s + 1])
        p = deg[a:b] / _total(deg[a:b], a, b)
        c = p.cumsum()
        out[a:b] = c / c[-1]
    return out


def max_nodes_table(max_degree: int, rw_hops: int, restart_prob: float) -> np.ndarray:
    """graph_dataset.py:113-124, tabulated by in-degree (L depends on nothing else)."""
    out = np.empty(max_degree + 1, dtype=np.int32)
    for d in range(max_degree + 1):
        out[d] = max(rw_hops, int((d ** 0.75) * math.e / (math.e - 1) / restart_prob + 0.5))
    return out


def restart_threshold(restart_prob: float) -> int:
    return min(int(restart_prob * 4294967296.0), MASK32)


# --------------------------------------------------------------------------
# pure-Python restatement
# --------------------------------------------------------------------------
def py_philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK32, p1 & MASK32, \
                         ((p0 >> 32) ^ c3 ^ k1) & MASK32, p0 & MASK32
        k0 = (k0 + 0x9E3779B9) & MASK32
        k1 = (k1 + 0xBB67AE85) & MASK32
    return [c0, c1, c2, c3]


def py_draw_seed(cdf, run_seed: int, sample_id: int, shard_off=None, batch_size: int = 1) -> int:
    key = [(run_seed & MASK32) ^ 0x5EED5EED, ((run_seed >> 32) & MASK32) ^ 0x00A11CE5]
    x = py_philox4x32_10([sample_id & MASK32, (sample_id >> 32) & MASK32, 0, 0], key)
    u = ((x[0] << 21) | (x[1] >> 11)) / 9007199254740992.0
    lo, hi = 0, len(cdf)
    if shard_off is not None and len(shard_off) > 2:       # the shard of this sample's DataLoader batch
        sh = (sample_id // batch_size) % (len(shard_off) - 1)
        lo, hi = int(shard_off[sh]), int(shard_off[sh + 1])
    idx = lo + int(np.searchsorted(cdf[lo:hi], u, side="right"))
    return min(idx, hi - 1)


def py_rwr_trace(row_ptr, col_idx, seed: int, L: int, run_seed: int, g: int, restart_u32: int):
    """DGL random_walk_with_restart semantics (graph_dataset.py:125-130), one view."""
    key = [run_seed & MASK32, (run_seed >> 32) & MASK32]

    def word(walk, i):
        return py_philox4x32_10([walk, i >> 2, g & MASK32, (g >> 32) & MASK32], key)[i & 3]

    trace, walk = [], 0
    while len(trace) < L:
        cur, t = seed, 0
        while True:
            if t > 0 and word(walk, 2 * t - 1) < restart_u32:
                break
            beg, end = int(row_ptr[cur]), int(row_ptr[cur + 1])
            cur = int(col_idx[beg + ((word(walk, 2 * t) * (end - beg)) >> 32)])
            trace.append(cur)
            if len(trace) == L:
                break
            t += 1
        walk += 1
    return trace


def py_subgraph(row_ptr, col_idx, seed: int, trace):
    """data_util.py:221-230: node list, then induced CSR with local ids."""
    nodes = [seed] + sorted(set(trace) - {seed})
    local = {v: i for i, v in enumerate(nodes)}
    rp, col = [0], []
    for v in nodes:
        for e in range(int(row_ptr[v]), int(row_ptr[v + 1])):
            u = int(col_idx[e])
            if u in local:
                col.append(local[u])
        rp.append(len(col))
    return nodes, rp, col


# --------------------------------------------------------------------------
# C oracle binding
# --------------------------------------------------------------------------
def _build():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "sampler_oracle.c")):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class COracle:
    def __init__(self):
        _build()
        lib = ctypes.CDLL(_LIB)
        lib.oracle_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        lib.oracle_philox4x32_10.restype = None
        lib.oracle_draw_seeds.argtypes = [_f64p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64,
                                          ctypes.c_int32, _i32p]
        lib.oracle_draw_seeds.restype = None
        lib.oracle_draw_seeds_sharded.argtypes = [_f64p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_uint64, ctypes.c_int64, ctypes.c_int32, _i32p]
        lib.oracle_draw_seeds_sharded.restype = None
        lib.oracle_rwr_trace.argtypes = [_i32p, _i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64,
                                         ctypes.c_uint64, ctypes.c_uint32, _i32p]
        lib.oracle_rwr_trace.restype = ctypes.c_int32
        lib.oracle_node_set.argtypes = [_i32p, ctypes.c_int32, ctypes.c_int32, _i32p]
        lib.oracle_node_set.restype = ctypes.c_int32
        lib.oracle_sample_batch.argtypes = [
            _i32p, _i32p, ctypes.c_int64, _i32p, _i32p, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_uint64, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_int64, ctypes.c_int64, _i32p, _i64p, _i32p, _i32p, _i32p, _i64p]
        lib.oracle_sample_batch.restype = ctypes.c_int32
        lib.oracle_max_threads.restype = ctypes.c_int32
        self.lib = lib

    def philox(self, ctr, key):
        out = np.zeros(4, dtype=np.uint32)
        self.lib.oracle_philox4x32_10(np.asarray(ctr, dtype=np.uint32), np.asarray(key, dtype=np.uint32), out)
        return out

    def draw_seeds(self, cdf, run_seed, first_sample_id, count, shard_off=None, batch_size=None):
        seeds = np.empty(count, dtype=np.int32)
        if shard_off is not None and len(shard_off) > 2:
            so = np.ascontiguousarray(shard_off, dtype=np.int64)
            self.lib.oracle_draw_seeds_sharded(np.ascontiguousarray(cdf), len(cdf), so.ctypes.data, len(so) - 1,
                                               int(batch_size or count), run_seed, first_sample_id, count, seeds)
            return seeds
        self.lib.oracle_draw_seeds(np.ascontiguousarray(cdf), len(cdf), run_seed, first_sample_id, count, seeds)
        return seeds

    def rwr_trace(self, row_ptr, col_idx, seed, L, run_seed, g, restart_u32):
        trace = np.empty(L, dtype=np.int32)
        self.lib.oracle_rwr_trace(row_ptr, col_idx, seed, L, run_seed, g, restart_u32, trace)
        return trace

    def max_threads(self):
        return int(self.lib.oracle_max_threads())

    def sample_batch(self, row_ptr, col_idx, seeds, L, view, run_seed, first_sample_id, restart_u32,
                     clear_visit_counts=False, threads=1):
        """-> dict(node_off, edge_off, parent_nid, row_ptr, col_idx, steps, scanned_edges)."""
        B = len(seeds)
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        L = np.ascontiguousarray(L, dtype=np.int32)
        node_cap = int(L.sum()) + B
        edge_cap = max(1 << 16, 64 * node_cap)
        while True:
            node_off = np.zeros(B + 1, dtype=np.int32)
            edge_off = np.zeros(B + 1, dtype=np.int64)
            parent = np.empty(node_cap, dtype=np.int32)
            rp = np.empty(node_cap + 1, dtype=np.int32)
            col = np.empty(edge_cap, dtype=np.int32)
            stats = np.zeros(2, dtype=np.int64)
            rc = self.lib.oracle_sample_batch(
                row_ptr, col_idx, len(row_ptr) - 1, seeds, L, B, view, run_seed, first_sample_id,
                restart_u32, int(clear_visit_counts), threads, node_cap, edge_cap,
                node_off, edge_off, parent, rp, col, stats)
            if rc == 0:
                break
            edge_cap = max(edge_cap * 2, int(edge_off[B]))
        N, nnz = int(node_off[B]), int(edge_off[B])
        return dict(node_off=node_off, edge_off=edge_off, parent_nid=parent[:N].copy(),
                    row_ptr=rp[:N + 1].copy(), col_idx=col[:nnz].copy(),
                    steps=int(stats[0]), scanned_edges=int(stats[1]))
