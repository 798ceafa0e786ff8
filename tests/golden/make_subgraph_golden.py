"""Generates tests/golden/subgraph_nodes_reference.json by EXECUTING THE REFERENCE'S OWN ``_rwr_trace_to_dgl_graph``
(/root/reference/gcc/datasets/data_util.py:218-239) on random-walk traces: which nodes a trace selects and in which order
(``torch.unique`` ascending, the seed taken out and put first) and where the seed flag goes.  DGL is replaced by
tests/golden/dgl_stub.py; the parent graph is a stand-in whose ``subgraph(nodes)`` records the node list it is given and
returns the induced subgraph (so that the reference's positional-embedding call that follows has something to work on).
What ``g.subgraph`` itself does inside DGL (edge order of the induced subgraph) stays DGL-recalled -- this pins the part that
is the reference's own Python.  Run from the repo root:  python tests/golden/make_subgraph_golden.py
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
backend = types.ModuleType("dgl.backend")
backend.asnumpy = lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
sys.modules["dgl.backend"] = backend
sys.modules["dgl"].backend = backend
sys.path.insert(0, "/root/reference")

from gcc.datasets import data_util  # noqa: E402
from make_posemb_golden import StubGraph  # noqa: E402

from gcc_amd.graphgen import powerlaw_graph  # noqa: E402
from oracle import sampler as O  # noqa: E402


class Parent:
    def __init__(self, rp, ci):
        self.rp, self.ci, self.asked = rp, ci, None

    def subgraph(self, nodes):
        self.asked = [int(v) for v in nodes]
        _, lrp, lci = O.py_subgraph(self.rp, self.ci, self.asked[0], self.asked[1:])
        return StubGraph(lrp, lci)


def main():
    rp, ci = powerlaw_graph(3000, 30000, 3)
    rng = np.random.default_rng(11)
    items = []
    for i in range(12):
        seed = int(rng.integers(0, len(rp) - 1))
        L = int(rng.integers(4, 120))
        trace = O.py_rwr_trace(rp, ci, seed, L, run_seed=9, g=i, restart_u32=O.restart_threshold(0.8))
        if i % 3 == 0:
            trace = trace + [seed]                           # the walk came back to its seed
        cuts = sorted(set(int(c) for c in rng.integers(1, len(trace), 3)))
        pieces = [torch.tensor(trace[a:b], dtype=torch.long) for a, b in zip([0] + cuts, cuts + [len(trace)])]   # dgl returns one tensor per restart
        g = Parent(rp, ci)
        np.random.seed(i)
        sub = data_util._rwr_trace_to_dgl_graph(g, seed, pieces, 32)
        flag = sub.ndata["seed"].tolist()
        items.append(dict(seed=seed, trace=trace, nodes=g.asked, seed_flag_at=[j for j, f in enumerate(flag) if f]))
        print(i, "seed", seed, "trace", len(trace), "nodes", len(g.asked), "flag", items[-1]["seed_flag_at"])
    with open(os.path.join(HERE, "subgraph_nodes_reference.json"), "w") as f:
        json.dump(dict(graph="powerlaw_graph(3000, 30000, 3)", items=items), f)


if __name__ == "__main__":
    main()
