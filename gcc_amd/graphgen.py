"""Synthetic power-law pre-training graphs (SURVEY.md §8d "Synthetic inputs").

The reference trains on ``data/small.bin`` which cannot be downloaded here, so
bench.py and the parity tests use a deterministic Chung-Lu graph that obeys the
same *input contract* the reference's converter establishes
(/root/reference/gcc/utils/x2dgl.py:39-62): symmetric, no self loops, no
duplicate edges, no zero-degree nodes, immutable CSR.  Rows are sorted
ascending and ids are int32.

Nothing here is on the timed path; it only produces inputs.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sp

__all__ = ["powerlaw_graph", "tiny_graphs", "check_contract"]


def _expected_degrees(num_nodes: int, target_sum: float, gamma: float, wmax: float) -> np.ndarray:
    # inverse CDF of a Pareto(gamma) truncated to [1, wmax], evaluated on a
    # fixed lattice so that the sequence does not depend on an RNG
    u = (np.arange(num_nodes, dtype=np.float64) + 0.5) / num_nodes
    a = 1.0 - gamma
    base = (1.0 - u * (1.0 - wmax ** a)) ** (1.0 / a)
    lo, hi = 1e-3, 1e3
    for _ in range(80):  # bisection on the scale so that sum(w) == target_sum
        mid = 0.5 * (lo + hi)
        s = np.clip(base * mid, 1.0, wmax).sum()
        if s < target_sum:
            lo = mid
        else:
            hi = mid
    return np.clip(base * (0.5 * (lo + hi)), 1.0, wmax)


def powerlaw_graph(num_nodes: int, num_directed_edges: int, seed: int = 0,
                   gamma: float = 2.2):
    """Chung-Lu graph -> (row_ptr int32[V+1], col_idx int32[E]).

    ``num_directed_edges`` is the target for the symmetrised edge count
    (G1: 1_000_000 / 10_000_000, G2: 10_000_000 / 200_000_000); the realised
    count is slightly lower after duplicate / self-loop removal and V shrinks
    by the zero-degree nodes that are dropped (x2dgl.py:61).
    """
    cache = os.environ.get("GCC_AMD_GRAPH_CACHE")          # optional: a directory; the 10M/200M graph takes 3 minutes to build
    path = os.path.join(cache, f"powerlaw_{num_nodes}_{num_directed_edges}_{seed}_{gamma}.npz") if cache else None
    if path and os.path.exists(path):
        z = np.load(path)
        return z["row_ptr"], z["col_idx"]
    rp, ci = _powerlaw_graph(num_nodes, num_directed_edges, seed, gamma)
    if path:
        os.makedirs(cache, exist_ok=True)
        np.savez(path, row_ptr=rp, col_idx=ci)
    return rp, ci


def _chunked(fn, n, threads=None):
    """fn(lo, hi) over [0, n) in slices on a thread pool (numpy releases the GIL inside searchsorted / take / sort):
    the 10M/200M graph is 200M binary searches in an 80 MB table -- minutes on one core."""
    from concurrent.futures import ThreadPoolExecutor

    threads = threads or min(32, os.cpu_count() or 1)
    cuts = np.linspace(0, n, max(1, min(threads, n // (1 << 20)) if n >= (1 << 21) else 1) + 1).astype(np.int64)
    if len(cuts) == 2:
        return [fn(0, n)]
    with ThreadPoolExecutor(len(cuts) - 1) as ex:
        return list(ex.map(lambda ab: fn(int(ab[0]), int(ab[1])), zip(cuts[:-1], cuts[1:])))


def _powerlaw_graph(num_nodes, num_directed_edges, seed, gamma):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_und = num_directed_edges // 2
    wmax = 4.0 * np.sqrt(num_nodes)
    w = _expected_degrees(num_nodes, 2.0 * n_und, gamma, wmax)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    # scatter the heavy nodes over the id range so that id order carries no
    # degree information (real graphs are not degree-sorted)
    perm = rng.permutation(num_nodes)

    def draw(u):          # one RNG stream, drawn in order; the look-ups are sliced over threads
        return np.concatenate(_chunked(lambda a, b: perm[np.minimum(np.searchsorted(cdf, u[a:b], side="right"),
                                                                    num_nodes - 1)], len(u)))
    src = draw(rng.random(n_und))
    dst = draw(rng.random(n_und))
    keep = src != dst                                   # x2dgl.py:41-42
    src, dst = src[keep], dst[keep]
    lo = np.minimum(src, dst).astype(np.int64)
    hi = np.maximum(src, dst).astype(np.int64)
    key = np.unique(lo * num_nodes + hi)                # x2dgl.py:52-54 (dedup)
    lo, hi = key // num_nodes, key % num_nodes
    deg = np.bincount(lo, minlength=num_nodes) + np.bincount(hi, minlength=num_nodes)
    alive = deg > 0                                     # x2dgl.py:61
    relabel = np.cumsum(alive) - 1
    v = int(alive.sum())
    lo, hi = relabel[lo], relabel[hi]
    # both directions (x2dgl.py:43-47) as row-major keys row * v + col: the forward half is already in order (relabel is
    # monotone), the backward half is sorted on its own and a stable sort merges the two runs -- the sorted CSR of an
    # edge set is unique, so this is the matrix scipy's COO -> CSR + sort_indices gave, built in a third of the time
    back = hi * v + lo
    back.sort()
    keys = np.concatenate([lo * v + hi, back])
    del lo, hi, back, key
    keys.sort(kind="stable")
    rows = np.concatenate(_chunked(lambda a, b: keys[a:b] // v, len(keys)))
    cols = np.concatenate(_chunked(lambda a, b: (keys[a:b] - rows[a:b] * v).astype(np.int32), len(keys)))
    row_ptr = np.zeros(v + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=v), out=row_ptr[1:])
    return row_ptr.astype(np.int32), cols


def from_edges(num_nodes: int, edges) -> tuple[np.ndarray, np.ndarray]:
    """Undirected edge list -> symmetric sorted CSR (no relabelling)."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    rows = np.concatenate([e[:, 0], e[:, 1]])
    cols = np.concatenate([e[:, 1], e[:, 0]])
    a = sp.csr_matrix((np.ones(rows.shape[0], dtype=np.int8), (rows, cols)),
                      shape=(num_nodes, num_nodes))
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def tiny_graphs() -> dict:
    """Hand-checkable graphs used by the known-answer tests."""
    out = {}
    out["path5"] = from_edges(5, [(0, 1), (1, 2), (2, 3), (3, 4)])
    out["star6"] = from_edges(6, [(0, i) for i in range(1, 6)])
    out["tri_tail"] = from_edges(5, [(0, 1), (1, 2), (0, 2), (2, 3), (3, 4)])
    out["k4"] = from_edges(4, [(i, j) for i in range(4) for j in range(i + 1, 4)])
    return out


def check_contract(row_ptr: np.ndarray, col_idx: np.ndarray) -> None:
    """Raise ValueError unless the CSR obeys the sampler's input contract."""
    v = row_ptr.shape[0] - 1
    if row_ptr[0] != 0 or row_ptr[-1] != col_idx.shape[0]:
        raise ValueError("row_ptr does not span col_idx")
    deg = np.diff(row_ptr)
    if (deg <= 0).any():
        raise ValueError("zero-degree node: the reference removes them (x2dgl.py:61) "
                         "and DGL's walker aborts on them")
    if col_idx.min() < 0 or col_idx.max() >= v:
        raise ValueError("col_idx out of range")
    a = sp.csr_matrix((np.ones(col_idx.shape[0], dtype=np.int8), col_idx, row_ptr), shape=(v, v))
    if not a.has_sorted_indices:
        raise ValueError("rows must be sorted ascending")
    if a.diagonal().any():
        raise ValueError("self loop present (x2dgl.py:41-42)")
    b = a.copy()
    b.sum_duplicates()
    if b.nnz != a.nnz:
        raise ValueError("duplicate edge present (x2dgl.py:52-54)")
    if (a != a.T).nnz != 0:
        raise ValueError("graph is not symmetric (x2dgl.py:43-47)")
