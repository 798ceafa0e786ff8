"""CPU oracle of the positional embedding -- TEST INFRASTRUCTURE ONLY.

Restates /root/reference/gcc/datasets/data_util.py:242-281
(``_add_undirected_graph_positional_embedding`` + ``eigen_decomposision``):
top-k eigenvectors (k = min(n-2, hidden), which="LA") of D^-1/2 A D^-1/2 in
float64 through SciPy's ARPACK ``eigsh`` with ncv = min(n, max(2k+1, 20)) and a
random start vector, rows L2-normalised, cast to float32, zero-padded to
``hidden`` columns.

Parity status: pinned.  tests/golden/posemb_reference.npz holds outputs of the
reference's own function (tests/golden/make_posemb_golden.py executes it with
DGL stubbed, NumPy's global generator seeded right before each call, since the
reference seeds ARPACK with np.random.rand); tests/test_posemb_oracle_golden.py
requires this restatement to reproduce them under the same seed: element by
element where the wanted eigenvalues are simple, row norms and Gram matrix of the
rows where they repeat (ARPACK's basis inside a repeated eigenvalue depends on
rounding, e.g. on the BLAS thread count).  Eigenvectors are defined up to sign (and up to rotation inside degenerate
eigenspaces) and depend on the start vector, so the DEVICE solver is compared
with this oracle / dense float64 eigh on invariants (residual, eigenvalues,
subspace, row norms), not element-wise.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse import linalg


def normalized_adjacency(row_ptr, col_idx):
    """data_util.py:273-277: norm * adj * norm with norm = diag(clip(in_deg, 1)^-0.5)."""
    n = len(row_ptr) - 1
    adj = sp.csr_matrix((np.ones(len(col_idx), dtype=float), col_idx, row_ptr), shape=(n, n))
    in_deg = np.asarray(adj.sum(axis=0)).ravel()
    norm = sp.diags(np.clip(in_deg, 1, None) ** -0.5, dtype=float)
    return norm * adj * norm


def eigen_decomposition(n, k, laplacian, hidden_size, retry=10, rng=np.random):
    """data_util.py:242-263 -> (x[n, hidden] float32, eigenvalues[k] or None)."""
    if k <= 0:
        return np.zeros((n, hidden_size), dtype=np.float32), None
    laplacian = laplacian.astype("float64")
    ncv = min(n, max(2 * k + 1, 20))
    v0 = rng.rand(n).astype("float64")
    s, u = None, None
    for i in range(retry):
        try:
            s, u = linalg.eigsh(laplacian, k=k, which="LA", ncv=ncv, v0=v0)
        except linalg.ArpackError:       # scipy >= 1.8 name of sparse.linalg.eigen.arpack.ArpackError
            ncv = min(ncv * 2, n)
            if i + 1 == retry:
                u = np.zeros((n, k))
        else:
            break
    norms = np.linalg.norm(u, axis=1, keepdims=True)          # sklearn normalize(norm="l2")
    x = u / np.where(norms == 0, 1.0, norms)
    x = x.astype("float32")
    x = np.pad(x, ((0, 0), (0, hidden_size - k)), "constant")
    return x, s


def positional_embedding(row_ptr, col_idx, hidden_size=32, rng=np.random):
    """One subgraph (local CSR) -> pos_undirected [n, hidden] float32 (data_util.py:266-281)."""
    n = len(row_ptr) - 1
    lap = normalized_adjacency(np.asarray(row_ptr), np.asarray(col_idx))
    k = min(n - 2, hidden_size)
    return eigen_decomposition(n, k, lap, hidden_size, rng=rng)


def batched_positional_embedding(node_off, row_ptr, col_idx, hidden_size=32, seed=0):
    """Applies :func:`positional_embedding` to every block of a batched CSR."""
    rng = np.random.RandomState(seed)
    node_off = np.asarray(node_off)
    out = np.zeros((int(node_off[-1]), hidden_size), dtype=np.float32)
    for b in range(len(node_off) - 1):
        lo, hi = int(node_off[b]), int(node_off[b + 1])
        rp = np.asarray(row_ptr[lo:hi + 1]) - row_ptr[lo]
        ci = np.asarray(col_idx[row_ptr[lo]:row_ptr[hi]]) - lo
        out[lo:hi], _ = positional_embedding(rp, ci, hidden_size, rng)
    return out
