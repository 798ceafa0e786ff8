"""Generates tests/golden/shard_reference.json by EXECUTING THE REFERENCE'S OWN ``LoadBalanceGraphDataset.__init__`` (the greedy
load balancing of graphs over workers, graph_dataset.py:33-80) and ``__iter__`` (per-worker seed distribution
p(v) = in_degree(v) ** 0.75 / sum over the worker's graphs, graph_dataset.py:84-92), with ``dgl.data.utils.load_labels``
returning the corpus' graph sizes and ``np.random.choice`` replaced by a recorder of its ``p`` argument (the draw itself is the
host generator's and is not what the device reproduces: the device draws from the same distribution with Philox).  The corpus is
``tests/shard_check.py: corpus()``.  Run from the repo root:  python tests/golden/make_shard_golden.py
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

from tests.shard_check import corpus  # noqa: E402  (before /root/reference, which has a tests package of its own, goes on the path)

sys.path.insert(0, "/root/reference")
GRAPHS = corpus()
utils = types.ModuleType("dgl.data.utils")
utils.load_labels = lambda path: {"graph_sizes": torch.tensor([len(rp) - 1 for rp, _ in GRAPHS])}
sys.modules["dgl.data.utils"] = utils
dgl.data.utils = utils

from gcc.datasets import graph_dataset  # noqa: E402


class G:
    def __init__(self, rp, ci):
        self.rp, self.ci = np.asarray(rp), np.asarray(ci)

    def number_of_nodes(self):
        return len(self.rp) - 1

    def in_degrees(self):
        return torch.from_numpy(np.bincount(self.ci, minlength=self.number_of_nodes()))


def main():
    out = []
    for num_workers, num_copies in ((1, 1), (2, 1), (3, 1), (4, 2)):
        ds = graph_dataset.LoadBalanceGraphDataset(num_workers=num_workers, num_copies=num_copies, num_samples=5,
                                                   dgl_graphs_file="unused.bin")
        shards = []
        for w in range(num_workers):
            ds.graphs = [G(*GRAPHS[i]) for i in ds.jobs[w]]           # what worker_init_fn loads for worker w
            ds.length = sum(g.number_of_nodes() for g in ds.graphs)
            rec = {}
            real = np.random.choice

            def choice(n, size=None, replace=True, p=None):
                rec.update(n=int(n), size=int(size), p=np.asarray(p, dtype=np.float64).copy())
                return np.zeros(0, dtype=np.int64)                    # no samples: __getitem__ is not what is recorded here

            np.random.choice = choice
            try:
                list(iter(ds))
            finally:
                np.random.choice = real
            assert rec["n"] == ds.length and abs(rec["p"].sum() - 1) < 1e-12
            shards.append(dict(jobs=[int(i) for i in ds.jobs[w]], length=rec["n"], p_head=rec["p"][:64].tolist(),
                               p_sum_sq=float((rec["p"] ** 2).sum()), p_argmax=int(rec["p"].argmax()), p_max=float(rec["p"].max())))
        out.append(dict(num_workers=num_workers, num_copies=num_copies, total=int(ds.total), jobs=[[int(i) for i in j] for j in ds.jobs],
                        shards=shards))
        print(num_workers, num_copies, ds.jobs)
    json.dump(dict(corpus="tests/shard_check.py: corpus()", configs=out), open(os.path.join(HERE, "shard_reference.json"), "w"))


if __name__ == "__main__":
    main()
