"""The GraphDataset family used by the reference's ``generate.py`` (gcc/datasets/graph_dataset.py:180-340), on the
device sampler: one item per NODE of the graph, in node order, both views from the same seed
(``step_dist = [1, 0, 0]``), ``max_nodes_per_seed`` from the out-degree without the 0.75 power (:244-255).

Yields already-batched ``(graph_q, graph_k)`` pairs like ``gcc_amd.sampler.LoadBalanceGraphDataset``; the last batch
is padded with node 0 and reports ``valid`` rows.  Graph classification (``entire_graph=True`` over a list of small
graphs) is not part of this path yet."""
from __future__ import annotations

import numpy as np

from .graph import max_nodes_out_degree_table


class NodeClassificationDataset:
    def __init__(self, dataset=None, rw_hops=64, subgraph_size=64, restart_prob=0.8, positional_embedding_size=32,
                 step_dist=(1.0, 0.0, 0.0), graph=None, edge_multiplicity=2, batch_size=256, run_seed=0,
                 device="cuda", sample_fn=None):
        """``graph`` = (row_ptr, col_idx) of the SIMPLE symmetric graph; ``edge_multiplicity`` = copies of every edge in
        the reference's DGL graph (gcc_amd.ingest.read_edgelist reports it).  ``sample_fn(first_id, seeds) -> (q, k)``
        is injectable for the emulator tests."""
        if list(step_dist) != [1.0, 0.0, 0.0]:
            raise NotImplementedError("step_dist other than [1, 0, 0] (generate.py and train.py never pass one)")
        assert positional_embedding_size > 1                       # graph_dataset.py:290
        if graph is None:
            raise ValueError("pass graph=(row_ptr, col_idx); named datasets need their files (gcc_amd.ingest)")
        self.dataset = dataset
        self.rw_hops, self.subgraph_size, self.restart_prob = rw_hops, subgraph_size, restart_prob
        self.positional_embedding_size = positional_embedding_size
        self.step_dist = list(step_dist)
        self.edge_multiplicity = int(edge_multiplicity)
        self.batch_size = int(batch_size)
        row_ptr, col_idx = graph
        self.length = int(len(row_ptr) - 1)                        # one item per node, :293
        self.total = self.length
        self.ltab = max_nodes_out_degree_table(int(np.diff(row_ptr).max()), rw_hops, restart_prob, self.edge_multiplicity)
        self._sample = sample_fn
        if sample_fn is None:
            from .graph import DeviceGraph
            from .sampler import DeviceRWRSampler

            self.graph = DeviceGraph(row_ptr, col_idx, rw_hops=rw_hops, restart_prob=restart_prob, device=device,
                                     ltab=self.ltab)
            self.sampler = DeviceRWRSampler(self.graph, self.batch_size, run_seed=run_seed)
            self._sample = self._device_sample

    def _device_sample(self, first_id, seeds):
        import torch

        return self.sampler.sample(first_id, seeds=torch.from_numpy(seeds).to(self.graph.device))

    def __len__(self):
        return self.length

    def num_batches(self):
        return (self.length + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        B = self.batch_size
        for i in range(self.num_batches()):
            lo = i * B
            valid = min(B, self.length - lo)
            seeds = np.zeros(B, dtype=np.int32)
            seeds[:valid] = np.arange(lo, lo + valid, dtype=np.int32)
            q, k = self._sample(lo, seeds)
            for g in (q, k):
                g.edge_multiplicity = self.edge_multiplicity
                g.valid = valid
            yield q, k
