"""The GraphDataset family used by the reference's ``generate.py`` (gcc/datasets/graph_dataset.py:180-340), on the
device sampler: one item per NODE of the graph, in node order, both views from the same seed
(``step_dist = [1, 0, 0]``), ``max_nodes_per_seed`` from the out-degree without the 0.75 power (:244-255).

Yields already-batched ``(graph_q, graph_k)`` pairs like ``gcc_amd.sampler.LoadBalanceGraphDataset``; the last batch
is padded with node 0 and reports ``valid`` rows.

``GraphClassificationDataset`` (:306-330): one item per GRAPH of a list of small graphs, ``entire_graph=True``: the
"subgraph" is the whole graph in its own node order, the seed flag marks ``out_degrees().argmax()``
(data_util.py:228-237) and both views are identical (the random walk's result is discarded), so batches are
assembled without the sampler."""
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


class GraphClassificationDataset:
    def __init__(self, dataset=None, rw_hops=64, subgraph_size=64, restart_prob=0.8, positional_embedding_size=32,
                 step_dist=(1.0, 0.0, 0.0), graphs=None, edge_multiplicity=1, batch_size=256, device="cuda"):
        """``graphs``: list of (row_ptr, col_idx) of simple symmetric graphs (what TUDataset holds for
        imdb-binary / imdb-multi / rdt-b / rdt-5k / collab); the dataset files themselves are not bundled."""
        if list(step_dist) != [1.0, 0.0, 0.0]:
            raise NotImplementedError("step_dist other than [1, 0, 0]")
        assert positional_embedding_size > 1
        if graphs is None:
            raise ValueError("pass graphs=[(row_ptr, col_idx), ...]")
        self.dataset, self.entire_graph = dataset, True               # graph_dataset.py:320
        self.rw_hops, self.subgraph_size, self.restart_prob = rw_hops, subgraph_size, restart_prob
        self.positional_embedding_size = positional_embedding_size
        self.graphs = [(np.asarray(rp, dtype=np.int64), np.asarray(ci, dtype=np.int64)) for rp, ci in graphs]
        self.length = self.total = len(self.graphs)
        self.edge_multiplicity = int(edge_multiplicity)
        self.batch_size = int(batch_size)
        self.device = device
        # capacity convention of the pipeline (as DeviceRWRSampler: B * (largest subgraph + 1)): the eigensolver sizes its
        # per-subgraph workspace as node_cap / batch_size
        self.node_cap = self.batch_size * (max(len(rp) - 1 for rp, _ in self.graphs) + 1)

    def __len__(self):
        return self.length

    def _convert_idx(self, idx):                                      # :326-329
        rp, _ = self.graphs[idx]
        return idx, int(np.argmax(np.diff(rp)))

    def _batch(self, lo, hi):
        import torch

        from .sampler import BatchedCSR

        B = self.batch_size
        node_off, edge_off, rows, cols, seeds = [0], [0], [], [], []
        for idx in range(lo, hi):
            rp, ci = self.graphs[idx]
            o = node_off[-1]
            rows.append(rp[1:] + edge_off[-1])
            cols.append(ci + o)
            seeds.append(self._convert_idx(idx)[1])
            node_off.append(o + len(rp) - 1)
            edge_off.append(edge_off[-1] + len(ci))
        for _ in range(hi - lo, B):                                   # padding: empty graphs
            node_off.append(node_off[-1])
            edge_off.append(edge_off[-1])
            seeds.append(0)
        n, e = node_off[-1], edge_off[-1]
        i32 = dict(dtype=torch.int32, device=self.device)
        row_ptr = torch.zeros(self.node_cap + 1, **i32)
        row_ptr[: n + 1] = torch.from_numpy(np.concatenate([[0]] + rows).astype(np.int32)).to(self.device)
        graph_id = torch.zeros(self.node_cap, **i32)
        graph_id[:n] = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(np.diff(node_off))).to(self.device)
        g = BatchedCSR(B, torch.tensor(node_off, **i32), torch.tensor(edge_off, **i32), torch.zeros(self.node_cap, **i32),
                       graph_id, row_ptr, torch.from_numpy(np.concatenate(cols).astype(np.int32)).to(self.device))
        g.seed_local = torch.tensor(seeds, **i32)
        g.edge_multiplicity = self.edge_multiplicity
        g.valid = hi - lo
        return g

    def __iter__(self):
        for lo in range(0, self.length, self.batch_size):
            g = self._batch(lo, min(lo + self.batch_size, self.length))
            yield g, g                                                # graph_q and graph_k are the same whole graph
