"""Generates tests/golden/edgelist_reference.json by EXECUTING THE REFERENCE'S OWN ``data_util.Edgelist`` (edge-list + node-label
reader, data_util.py:62-109) and ``NodeClassificationDataset._create_dgl_graph`` (graph_dataset.py:300-308) on a small
edge-list file, with ``dgl.DGLGraph`` replaced by a recorder of ``add_nodes`` / ``add_edges``: node re-indexing by first
appearance, one-hot labels by first appearance, and the multigraph the reference walks on (every listed edge in both directions
by the reader, both again by ``_create_dgl_graph``: 2 copies per direction).  The file's text is stored with the outputs.
Run from the repo root:  python tests/golden/make_edgelist_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import dgl_stub  # noqa: E402

dgl_stub.install()
sys.path.insert(0, "/root/reference")


class RecordingGraph:
    def __init__(self):
        self.num_nodes, self.edges, self.is_readonly = 0, [], False

    def add_nodes(self, n):
        self.num_nodes += int(n)

    def add_edges(self, src, dst):
        self.edges += list(zip([int(s) for s in src], [int(d) for d in dst]))

    def readonly(self):
        self.is_readonly = True


sys.modules["dgl"].DGLGraph = RecordingGraph

from gcc.datasets import data_util, graph_dataset  # noqa: E402


def main():
    rng = np.random.RandomState(0)
    n, extra = 40, 50
    ids = rng.permutation(1000)[:n] + 5
    pairs = {(i, i + 1) for i in range(n - 1)}
    while len(pairs) < n - 1 + extra:
        a, b = sorted(rng.randint(0, n, 2))
        if a != b:
            pairs.add((a, b))
    pairs = sorted(pairs)
    rng.shuffle(pairs)                                       # first appearance is not sorted order
    edgelist = "\n".join(f"{ids[a]} {ids[b]}" for a, b in pairs) + "\n"
    labels = rng.randint(0, 3, n)
    nodelabel = "\n".join(f"{ids[i]} {labels[i] + 7}" for i in rng.permutation(n)) + "\n"
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "toy.edgelist"), "w").write(edgelist)
        open(os.path.join(td, "toy.nodelabel"), "w").write(nodelabel)
        ds = data_util.Edgelist(td, "toy")
    data = ds.get(0)
    g = graph_dataset.NodeClassificationDataset._create_dgl_graph(None, data)
    assert g.is_readonly
    out = dict(edgelist=edgelist, nodelabel=nodelabel, node2id={str(k): int(v) for k, v in ds.node2id.items()},
               y=data.y.long().tolist(), edge_index=data.edge_index.tolist(), num_nodes=int(g.num_nodes), dgl_edges=g.edges)
    json.dump(out, open(os.path.join(HERE, "edgelist_reference.json"), "w"))
    print("nodes", g.num_nodes, "listed edges", len(pairs), "edge_index columns", data.edge_index.shape[1], "DGL edges", len(g.edges))


if __name__ == "__main__":
    main()
