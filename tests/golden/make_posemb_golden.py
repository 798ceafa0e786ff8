"""Generates tests/golden/posemb_reference.npz by EXECUTING THE REFERENCE'S OWN
``_add_undirected_graph_positional_embedding`` / ``eigen_decomposision``
(/root/reference/gcc/datasets/data_util.py:242-281) in this container on a handful of small graphs, with DGL replaced by
tests/golden/dgl_stub.py plus a stand-in for the three DGLGraph members that function touches
(``number_of_nodes``, ``adjacency_matrix_scipy``, ``in_degrees``; ``dgl.backend.asnumpy``).  The start vector of ARPACK is
``np.random.rand(n)`` (data_util.py:248), so every call is made right after ``np.random.seed(seed)`` and the seed is stored:
with the same seed and the same SciPy the CPU oracle (oracle/posemb.py) must reproduce these arrays element by element.
/root/reference does not exist on the GPU box, so the vectors are committed.  Run from the repo root:

    python tests/golden/make_posemb_golden.py
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import dgl_stub  # noqa: E402

dgl_stub.install()
backend = types.ModuleType("dgl.backend")
backend.asnumpy = lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)     # data_util.py:275
sys.modules["dgl.backend"] = backend
sys.modules["dgl"].backend = backend
sys.path.insert(0, "/root/reference")

from gcc.datasets import data_util  # noqa: E402

from gcc_amd.graphgen import powerlaw_graph  # noqa: E402
from oracle import sampler as O  # noqa: E402

HID = 32


class StubGraph:
    """The members of a DGLGraph that data_util.py:266-281 uses, over a local CSR (row u lists its successors)."""

    def __init__(self, row_ptr, col_idx):
        self.rp, self.ci = np.asarray(row_ptr, dtype=np.int64), np.asarray(col_idx, dtype=np.int64)
        self.ndata = {}

    def number_of_nodes(self):
        return len(self.rp) - 1

    def adjacency_matrix_scipy(self, transpose=False, return_edge_ids=False):
        # dgl 0.4.x (DGL-recalled): transpose=False -> rows are destinations; the subgraphs here are symmetric, so the
        # orientation does not matter
        n = self.number_of_nodes()
        return sp.csr_matrix((np.ones(len(self.ci)), self.ci, self.rp), shape=(n, n))

    def in_degrees(self):
        return torch.from_numpy(np.bincount(self.ci, minlength=self.number_of_nodes()))


def sym_csr(n, edges):
    e = np.array(edges)
    a = sp.csr_matrix((np.ones(2 * len(e)), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int64), a.indices.astype(np.int64)


def graphs():
    rng = np.random.default_rng(7)
    out = [("triangle", *sym_csr(3, [(0, 1), (1, 2), (0, 2)])),                     # k = 1
           ("path4", *sym_csr(4, [(0, 1), (1, 2), (2, 3)])),                          # k = 2
           ("star9", *sym_csr(9, [(0, i) for i in range(1, 9)])),                     # twin leaves: repeated eigenvalue 0
           ("path40", *sym_csr(40, [(i, i + 1) for i in range(39)])),                 # k = 32 < n - 2
           ("random60", *sym_csr(60, [(i, i + 1) for i in range(59)]
                                 + [tuple(rng.integers(0, 60, 2)) for _ in range(120)]))]
    # (no self loops, no duplicates)
    rp60, ci60 = out[-1][1], out[-1][2]
    a = sp.csr_matrix((np.ones(len(ci60)), ci60, rp60), shape=(60, 60))
    a.setdiag(0)
    a.eliminate_zeros()
    a.data[:] = 1
    out[-1] = ("random60", a.indptr.astype(np.int64), a.indices.astype(np.int64))
    # two sampled ego-nets of the synthetic power-law graph (subgraphs of the oracle sampler, as the tests use them)
    rp, ci = powerlaw_graph(3000, 30000, 3)
    c = O.COracle()
    seeds = c.draw_seeds(O.seed_cdf(rp), 5, 0, 4)
    L = O.max_nodes_table(int(np.diff(rp).max()), 64, 0.8)[np.diff(rp)[seeds]]
    r = c.sample_batch(rp, ci, seeds, L, 0, 5, 0, O.restart_threshold(0.8))
    no = np.asarray(r["node_off"])
    for b in (0, 2):
        lo, hi = int(no[b]), int(no[b + 1])
        lrp = np.asarray(r["row_ptr"][lo:hi + 1]) - r["row_ptr"][lo]
        lci = np.asarray(r["col_idx"][r["row_ptr"][lo]:r["row_ptr"][hi]]) - lo
        out.append((f"egonet{b}", lrp.astype(np.int64), lci.astype(np.int64)))
    return out


def main():
    store = {}
    names = []
    for i, (name, rp, ci) in enumerate(graphs()):
        seed = 100 + i
        g = StubGraph(rp, ci)
        np.random.seed(seed)
        data_util._add_undirected_graph_positional_embedding(g, HID)
        x = g.ndata["pos_undirected"].numpy()
        assert x.shape == (len(rp) - 1, HID) and x.dtype == np.float32
        names.append(name)
        store[f"{name}_rp"], store[f"{name}_ci"], store[f"{name}_x"], store[f"{name}_seed"] = rp, ci, x, np.int64(seed)
        print(name, "n", len(rp) - 1, "nnz", len(ci), "k", min(len(rp) - 3, HID), "|x|", float(np.abs(x).sum()))
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "posemb_reference.npz"), **store)


if __name__ == "__main__":
    main()
