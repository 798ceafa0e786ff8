"""Graph files of the reference's downstream datasets -> int32 CSR.

``<name>.edgelist`` / ``<name>.nodelabel`` as read by ``Edgelist._preprocess`` (gcc/datasets/data_util.py:61-110):
one "u v" pair per line, node ids re-indexed in order of first appearance; "node label" per line.  The reference then
inserts every pair in both directions (data_util.py:84-85) and its dataset class inserts both directions AGAIN
(graph_dataset.py:301-302), so its DGL graph is a multigraph in which every undirected edge of the file exists
``2 x (times the pair is listed, in either order)`` times per direction.  The HIP path keeps a simple CSR plus ONE
uniform multiplicity (gcc_gin_pass.edge_multiplicity); files whose pairs repeat a non-uniform number of times, and self
loops, are rejected rather than approximated.

``read_tudataset``: the raw TU Dortmund collection layout (``<NAME>_A.txt``, ``<NAME>_graph_indicator.txt``,
``<NAME>_graph_labels.txt``) that DGL's ``TUDataset`` downloads for ``create_graph_classification_dataset``
(data_util.py:47-58: imdb-binary, imdb-multi, rdt-b, rdt-5k, collab) -> the list of small graphs
``GraphClassificationDataset(graphs=...)`` takes, plus the graph labels."""
from __future__ import annotations

import numpy as np


def csr_from_pairs(pairs: np.ndarray, num_nodes: int):
    """pairs int64 [m, 2] (undirected, any order, repeats allowed) -> (row_ptr, col_idx, multiplicity)."""
    if (pairs[:, 0] == pairs[:, 1]).any():
        raise ValueError("self loops are not supported by the sampler contract (x2dgl.py:41-42 removes them)")
    lo, hi = pairs.min(axis=1), pairs.max(axis=1)
    key, counts = np.unique(lo * num_nodes + hi, return_counts=True)
    if counts.min() != counts.max():
        raise ValueError("pairs repeat a non-uniform number of times: a general multigraph is not supported")
    u, v = key // num_nodes, key % num_nodes
    src = np.concatenate([u, v])
    dst = np.concatenate([v, u])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    if (np.diff(row_ptr) == 0).any():
        raise ValueError("isolated nodes (every node of an edge list has an edge; check the input)")
    return row_ptr.astype(np.int32), dst.astype(np.int32), int(counts[0])


def read_edgelist(edgelist_path: str, nodelabel_path: str = None, hindex: bool = False):
    """-> dict(row_ptr, col_idx, edge_multiplicity, node2id, y).  ``edge_multiplicity`` is the number of copies of
    every edge in the graph the reference's NodeClassificationDataset builds (2 x listings)."""
    node2id, pairs = {}, []
    with open(edgelist_path) as f:
        for line in f:
            if not line.strip():
                continue
            x, y = (int(t) for t in line.split()[:2])
            for n in (x, y):
                if n not in node2id:
                    node2id[n] = len(node2id)
            pairs.append((node2id[x], node2id[y]))
    num_nodes = len(node2id)
    row_ptr, col_idx, listed = csr_from_pairs(np.asarray(pairs, dtype=np.int64), num_nodes)
    out = dict(row_ptr=row_ptr, col_idx=col_idx, edge_multiplicity=2 * listed, node2id=node2id, y=None)
    if nodelabel_path is not None:
        nodes, labels, label2id = [], [], {}
        with open(nodelabel_path) as f:
            for line in f:
                if not line.strip():
                    continue
                x, label = (int(t) for t in line.split()[:2])
                if label not in label2id:
                    label2id[label] = len(label2id)
                nodes.append(node2id[x])
                labels.append(label if hindex else label2id[label])
        if hindex:                                   # data_util.py:104-106
            median = np.median(labels)
            labels = [int(l > median) for l in labels]
        y = np.zeros((num_nodes, max(len(label2id), 1)), dtype=np.float32)
        y[nodes, labels] = 1
        out["y"] = y
    return out


TU_NAMES = {"imdb-binary": "IMDB-BINARY", "imdb-multi": "IMDB-MULTI", "rdt-b": "REDDIT-BINARY",
            "rdt-5k": "REDDIT-MULTI-5K", "collab": "COLLAB"}                      # data_util.py:48-54


def read_tudataset(folder: str, name: str):
    """-> dict(graphs=[(row_ptr, col_idx), ...], graph_labels int64 [G], num_labels).  ``name`` is the reference's
    dataset name (``imdb-binary`` ...) or the TU name itself.  Nodes of a graph keep their file order (DGL builds each
    graph as the subgraph of its ascending node ids); labels are re-indexed 0..C-1 in ascending order of the file's
    values, as DGL's TUDataset does.  The adjacency must be symmetric without self loops or repeated entries."""
    import os

    tu = TU_NAMES.get(name, name)
    base = os.path.join(folder, tu + "_")
    edges = np.loadtxt(base + "A.txt", delimiter=",", dtype=np.int64, ndmin=2) - 1
    indicator = np.loadtxt(base + "graph_indicator.txt", dtype=np.int64, ndmin=1)
    labels = np.loadtxt(base + "graph_labels.txt", dtype=np.int64, ndmin=1)
    num_nodes = len(indicator)
    if (np.diff(indicator) < 0).any():
        raise ValueError("graph_indicator must be non-decreasing (nodes of a graph are contiguous in the TU format)")
    gids, first = np.unique(indicator, return_index=True)
    if len(gids) != len(labels):
        raise ValueError(f"{len(gids)} graphs in graph_indicator but {len(labels)} graph labels")
    src, dst = edges[:, 0], edges[:, 1]
    if len(src) and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= num_nodes):
        raise ValueError("node id out of range in A.txt")
    if (src == dst).any():
        raise ValueError("self loops are not supported by the sampler contract (x2dgl.py:41-42 removes them)")
    if (indicator[src] != indicator[dst]).any():
        raise ValueError("an edge connects two different graphs")
    key = src * num_nodes + dst
    if len(np.unique(key)) != len(key):
        raise ValueError("repeated adjacency entries: a general multigraph is not supported")
    if not np.array_equal(np.sort(key), np.sort(dst * num_nodes + src)):
        raise ValueError("the adjacency is not symmetric")
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    bounds = np.append(first, num_nodes)
    graphs = []
    for g in range(len(gids)):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        rp = row_ptr[lo:hi + 1] - row_ptr[lo]
        graphs.append((rp.astype(np.int32), (dst[row_ptr[lo]:row_ptr[hi]] - lo).astype(np.int32)))
    values = np.unique(labels)
    return dict(graphs=graphs, graph_labels=np.searchsorted(values, labels).astype(np.int64), num_labels=len(values))
