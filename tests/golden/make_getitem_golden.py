"""Generates tests/golden/getitem_calls_reference.json by EXECUTING THE REFERENCE'S OWN ``LoadBalanceGraphDataset.__getitem__``
(/root/reference/gcc/datasets/graph_dataset.py:94-179) with
``dgl.contrib.sampling.random_walk_with_restart`` replaced by a recorder: what the reference ASKS of DGL for a seed of a given
in-degree -- ``seeds=[v, v]`` (step_dist [1, 0, 0]: both views start at the same node), ``restart_prob`` and
``max_nodes_per_seed = max(rw_hops, int(deg ** 0.75 * e / (e - 1) / restart_prob + 0.5))``.  The object is made with
``object.__new__`` and given the attributes the method reads (its ``__init__`` loads a DGL file).  The walk itself stays inside
DGL and stays unpinned.  Run from the repo root:  python tests/golden/make_getitem_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import dgl_stub  # noqa: E402

dgl_stub.install()
dgl = sys.modules["dgl"]
backend = types.ModuleType("dgl.backend")
backend.asnumpy = lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
sys.modules["dgl.backend"] = backend
dgl.backend = backend
calls = []


def rwr(g, seeds, restart_prob, max_nodes_per_seed):
    calls.append(dict(seeds=[int(s) for s in seeds], restart_prob=float(restart_prob), max_nodes_per_seed=int(max_nodes_per_seed)))
    return [[torch.tensor([g.nbr(int(s))])] for s in seeds]        # one trace per seed: a single step to a neighbour


sampling = types.ModuleType("dgl.contrib.sampling")
sampling.random_walk_with_restart = rwr
contrib = types.ModuleType("dgl.contrib")
contrib.sampling = sampling
sys.modules["dgl.contrib"], sys.modules["dgl.contrib.sampling"] = contrib, sampling
dgl.contrib = contrib
sys.path.insert(0, "/root/reference")

from gcc.datasets import graph_dataset  # noqa: E402
from make_posemb_golden import StubGraph  # noqa: E402


class StarForest:
    """Parent graph: node d (d = 1 .. D) is the centre of a star with d leaves, so in_degree(d) = d; node 0 is isolated from
    the centres (a leaf of star 1)."""

    def __init__(self, D):
        self.D = D
        self.first_leaf = {d: D + 1 + d * (d - 1) // 2 for d in range(1, D + 1)}
        self.n = D + 1 + D * (D + 1) // 2

    def number_of_nodes(self):
        return self.n

    def in_degree(self, v):
        return v if 1 <= v <= self.D else 1

    def out_degree(self, v):                                 # (symmetric parent)
        return self.in_degree(v)

    def nbr(self, v):
        return self.first_leaf[v]

    def subgraph(self, nodes):
        nodes = [int(v) for v in nodes]
        assert nodes[0] in self.first_leaf and nodes[1:] == [self.first_leaf[nodes[0]]]
        return StubGraph([0, 1, 2], [1, 0])                 # centre - leaf


def main():
    items = []
    for rw_hops, restart_prob in ((256, 0.8), (64, 0.8), (16, 0.5), (4, 0.05)):
        ds = object.__new__(graph_dataset.LoadBalanceGraphDataset)
        ds.graphs = [StarForest(600)]
        ds.step_dist = [1.0, 0.0, 0.0]
        ds.aug = "rwr"
        ds.rw_hops, ds.restart_prob = rw_hops, restart_prob
        ds.positional_embedding_size = 32
        ds.graph_transform = None
        for deg in (1, 2, 3, 10, 47, 100, 255, 256, 257, 511, 600):
            calls.clear()
            np.random.seed(deg)
            ds[deg]                                           # node id = its in-degree
            assert len(calls) == 1
            items.append(dict(rw_hops=rw_hops, in_degree=deg, **calls[0]))
    # the GraphDataset family (graph_dataset.py:230-275; NodeClassificationDataset / GraphClassificationDataset of generate.py
    # inherit it): the OUT-degree enters without the 0.75 power
    for rw_hops, restart_prob in ((64, 0.8), (16, 0.5)):
        ds = object.__new__(graph_dataset.GraphDataset)
        ds.graphs = [StarForest(600)]
        ds.step_dist = [1.0, 0.0, 0.0]
        ds.rw_hops, ds.restart_prob = rw_hops, restart_prob
        ds.positional_embedding_size = 32
        for deg in (1, 2, 3, 10, 47, 100, 255, 256, 257, 511, 600):
            calls.clear()
            np.random.seed(deg)
            ds[deg]
            assert len(calls) == 1
            items.append(dict(family="GraphDataset", rw_hops=rw_hops, out_degree=deg, **calls[0]))
    # GraphClassificationDataset (graph_dataset.py:311-345): the "seed" of a whole graph is out_degrees().argmax(), both views
    # are the ENTIRE graph in its own node order (g.subgraph(g.nodes())) and the seed flag sits on that node
    class Whole:
        def __init__(self, deg):
            self.deg = list(deg)
            self.asked = None

        def number_of_nodes(self):
            return len(self.deg)

        def out_degrees(self):
            return torch.tensor(self.deg)

        def out_degree(self, v):
            return self.deg[v]

        def nodes(self):
            return torch.arange(len(self.deg))

        def nbr(self, v):
            return (v + 1) % len(self.deg)

        def subgraph(self, nodes):
            self.asked = [int(v) for v in nodes]
            n = len(self.deg)                                # a ring: enough for the positional embedding that follows
            return StubGraph(np.arange(0, 2 * n + 1, 2), np.array([[(i - 1) % n, (i + 1) % n] for i in range(n)]).ravel())

    for deg in ([2, 5, 3, 5, 1, 4], [7, 1, 1, 1, 1, 1, 1, 1], [1, 2, 3, 4, 5, 6, 7, 8, 9]):
        ds = object.__new__(graph_dataset.GraphClassificationDataset)
        ds.graphs = [Whole(deg)]
        ds.step_dist = [1.0, 0.0, 0.0]
        ds.rw_hops, ds.restart_prob, ds.positional_embedding_size, ds.entire_graph = 8, 0.8, 4, True
        calls.clear()
        np.random.seed(0)
        gq, gk = ds[0]
        flag = gq.ndata["seed"].tolist()
        items.append(dict(family="GraphClassificationDataset", out_degrees=deg, subgraph_nodes=ds.graphs[0].asked,
                          seed_flag_at=[j for j, f in enumerate(flag) if f], **calls[0]))
    with open(os.path.join(HERE, "getitem_calls_reference.json"), "w") as f:
        json.dump(items, f)
    print(len(items), "calls recorded; e.g.", items[0], items[-1])


if __name__ == "__main__":
    main()
