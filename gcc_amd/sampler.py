"""Device-resident RWR ego-net sampler behind the reference's dataset API.

Mirrors /root/reference/gcc/datasets/graph_dataset.py:23-179
(``LoadBalanceGraphDataset``, ``worker_init_fn``) and
gcc/datasets/data_util.py:26-32 (``batcher``): iterating the dataset yields
``(graph_q, graph_k)`` pairs that ``train.py`` feeds to ``GraphEncoder``.  The
pairs are produced already batched, in HBM, by the HIP kernels of
gcc_amd/csrc/sampler.hip; no worker processes, no pickling, no H2D copy.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _cabi
from .graph import DeviceGraph


class BatchedCSR:
    """One view's batched subgraphs = the ``dgl.batch(...)`` result the reference
    passes around (data_util.py:26-32).  Duck-types the DGLGraph members the hot
    path touches: ``ndata``, ``in_degrees``, ``to``,
    ``batch_size``, ``number_of_nodes``, ``number_of_edges``, ``batch_num_nodes``.

    All tensors are capacity-sized device buffers; the live extents are
    ``node_off[batch_size]`` nodes and ``edge_off[batch_size]`` edges and stay on
    the device (kernels read them there), so building a batch never syncs.
    """

    def __init__(self, batch_size, node_off, edge_off, parent_nid, graph_id, row_ptr, col_idx):
        self.batch_size = int(batch_size)
        self.node_off = node_off
        self.edge_off = edge_off
        self.parent_nid = parent_nid
        self.graph_id = graph_id
        self.row_ptr = row_ptr
        self.col_idx = col_idx
        self.pos_undirected = None      # [node_cap, P] f32, filled by gcc_amd.posemb
        self.edge_multiplicity = 1      # every CSR edge counts this many times (multigraph parents, gcc_gin_pass)
        self._n = None
        self._e = None
        self.ndata = _NData(self)

    # ---- DGLGraph surface -------------------------------------------------
    def number_of_nodes(self) -> int:
        if self._n is None:
            self._n = int(self.node_off[self.batch_size].item())
        return self._n

    def number_of_edges(self) -> int:
        if self._e is None:
            self._e = int(self.edge_off[self.batch_size].item())
        return self._e

    @property
    def batch_num_nodes(self):
        off = self.node_off[: self.batch_size + 1].cpu().numpy()
        return np.diff(off).tolist()

    def in_degrees(self):
        n = self.number_of_nodes()
        rp = self.row_ptr[: n + 1].long()
        return (rp[1:] - rp[:-1]) * self.edge_multiplicity          # symmetric parent => in-degree == row length

    def to(self, device):
        return self

    # ---- views used by tests / host code -----------------------------------
    def csr_numpy(self):
        n, e = self.number_of_nodes(), self.number_of_edges()
        return dict(node_off=self.node_off.cpu().numpy(), edge_off=self.edge_off.cpu().numpy(),
                    parent_nid=self.parent_nid[:n].cpu().numpy(), graph_id=self.graph_id[:n].cpu().numpy(),
                    row_ptr=self.row_ptr[: n + 1].cpu().numpy(), col_idx=self.col_idx[:e].cpu().numpy())

    def c_struct(self) -> _cabi.GccBatchOut:
        return _cabi.GccBatchOut(
            node_off=self.node_off.data_ptr(), edge_off=self.edge_off.data_ptr(),
            parent_nid=self.parent_nid.data_ptr(), graph_id=self.graph_id.data_ptr(),
            row_ptr=self.row_ptr.data_ptr(), col_idx=self.col_idx.data_ptr(),
            node_cap=self.parent_nid.numel(), edge_cap=self.col_idx.numel())


class _NData:
    """``g.ndata[...]`` of the reference graphs (graph_encoder.py:153-162)."""

    def __init__(self, g: BatchedCSR):
        self._g = g

    def __getitem__(self, key):
        import torch

        g = self._g
        if key == "seed":                       # data_util.py:234-238: one-hot at local node 0
            n = g.number_of_nodes()
            s = torch.zeros(n, dtype=torch.long, device=g.node_off.device)
            first = g.node_off[: g.batch_size].long()
            local = getattr(g, "seed_local", None)
            if local is not None:                   # entire_graph=True: the seed keeps its own index, data_util.py:236-237
                keep = (g.node_off[1: g.batch_size + 1] > g.node_off[: g.batch_size])
                s[(first + local.long())[keep]] = 1
            else:
                s[first] = 1
            return s
        if key == "pos_undirected":
            if g.pos_undirected is None:
                raise KeyError("pos_undirected has not been computed for this batch")
            return g.pos_undirected[: g.number_of_nodes()]
        raise KeyError(key)

    def __contains__(self, key):
        return key == "seed" or (key == "pos_undirected" and self._g.pos_undirected is not None)


class DeviceRWRSampler:
    """Owns the output/workspace buffers and issues ``gcc_sample_batch``."""

    def __init__(self, graph: DeviceGraph, batch_size: int, run_seed: int = 0,
                 edge_cap: int | None = None, scratch_entries: int | None = None, num_buffers: int = 2,
                 max_steps: int = 1, hub_degree: int = 0, max_hubs: int = 0):
        """``max_steps``: most consecutive steps one call may cover (:meth:`sample_multi`; the workspace and the
        induction scratch are sized for that many batches, and the buffer ring must hold at least as many)."""
        import torch

        self.graph = graph
        self.lib = _cabi.load()
        # member rows of at least this parent degree are not scanned by the induction (gcc_sample_params.hub_degree:
        # 0 = the library's default of 512, < 0 = scan every row); the subgraphs are the same bit for bit
        self.hub_degree = int(hub_degree)
        self.max_hubs = int(max_hubs)         # most unscanned rows per subgraph (0 = the library's default and maximum, 32)
        self.batch_size = int(batch_size)
        self.run_seed = int(run_seed) & 0xFFFFFFFFFFFFFFFF
        dev = graph.device
        B = self.batch_size
        self.node_cap = B * (graph.lmax + 1)
        self.edge_cap = int(edge_cap) if edge_cap else max(64 * B * (graph.rw_hops + 1), 2 * (graph.lmax + 1) ** 2)
        # induction scratch (hit slots): induce_kernel reserves one 1024-entry slot per unit of 256 aligned quads of
        # the members' parent rows, i.e. ~ the SUM OF THE MEMBERS' PARENT DEGREES per subgraph (not min(deg, n) as the
        # first induction did) -- unbounded by n: a hub-only batch needs several times the average.  A walk visits nodes
        # in proportion to their degree, so a view scans about n * (size-biased mean degree) parent edges per subgraph
        # (n ~ rw_hops / 2.5); sized 3x that for both views (HBM is 288 GB).  An overflow never writes out of bounds: it
        # sets bit 0 of the device status word and leaves a truncated subgraph, which check_status() turns into an error
        # (train.py polls it at every log line, bench.py after the timed region).
        expected = int(3 * 2 * B * (graph.rw_hops / 2.5) * getattr(graph, "sb_degree", 0.0))
        # ``scratch_entries`` (argument) is PER STEP; a call covers up to ``max_steps`` steps, so the buffer handed to the
        # library holds ``self.scratch_entries`` = per-step entries x max_steps (``scratch_entries_per_step`` keeps the
        # argument's meaning; gcc_sample_multi's ``scratch_entries`` is the whole call's).
        self._default_scratch = max(32 << 20, 512 * B * (graph.rw_hops + 1), 8 * (graph.lmax + 1) ** 2, expected)
        self.scratch_entries_per_step = int(scratch_entries) if scratch_entries else self._default_scratch
        # steps per call: bounded by the library (GCC_SAMPLE_MAX_STEPS; one LDS word per subgraph in the prefix kernel,
        # (2 * B * steps + 1) * 4 + 512 <= 64 KiB)
        self.max_steps = max(1, min(int(max_steps), 16, 16255 // (2 * B), int(num_buffers)))
        self.scratch_entries = self.scratch_entries_per_step * self.max_steps
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.regrown = 0                   # times grow() enlarged the scratch / edge capacity after an overflow
        self._retired = []                 # replaced buffers stay alive: launches in flight on other streams may still use them
        self._snap = None                  # pinned host words of the status snapshots in flight (status_snapshot)
        self._snap_free = []
        self._alloc_workspace()
        i32 = dict(dtype=torch.int32, device=dev)
        self._ring = []
        for _ in range(num_buffers):
            views = []
            for _v in range(2):
                views.append(dict(
                    node_off=torch.zeros(B + 1, **i32), edge_off=torch.zeros(B + 1, **i32),
                    parent_nid=torch.zeros(self.node_cap, **i32), graph_id=torch.zeros(self.node_cap, **i32),
                    row_ptr=torch.zeros(self.node_cap + 1, **i32), col_idx=torch.zeros(self.edge_cap, **i32)))
            self._ring.append(views)
        self._next = 0

    def _alloc_workspace(self):
        import torch

        nbytes = self.lib.gcc_sampler_workspace_bytes_multi(self.graph.byref(), self.batch_size, self.max_steps,
                                                            self.scratch_entries)
        if nbytes < 0:
            raise RuntimeError(self.lib.gcc_last_error().decode())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.graph.device)

    def sample(self, first_sample_id: int, seeds=None, prof=None):
        """-> (BatchedCSR q, BatchedCSR k) for samples first_sample_id .. +B-1.

        ``seeds`` (int32 device tensor [B]) overrides the deg^0.75 seed draw.
        """
        import torch

        views = self._ring[self._next]
        self._next = (self._next + 1) % len(self._ring)
        q = BatchedCSR(self.batch_size, **views[0])
        k = BatchedCSR(self.batch_size, **views[1])
        params = _cabi.GccSampleParams(
            run_seed=self.run_seed, first_sample_id=int(first_sample_id), batch_size=self.batch_size,
            restart_u32=self.graph.restart_u32,
            seeds=_cabi.dev_ptr(seeds, torch.int32) if seeds is not None else None,
            prof=prof.handle if prof is not None else None, hub_degree=self.hub_degree, max_hubs=self.max_hubs)
        cq, ck = q.c_struct(), k.c_struct()
        rc = self.lib.gcc_sample_batch(
            self.graph.byref(), ctypes.byref(params), ctypes.byref(cq), ctypes.byref(ck),
            self.workspace.data_ptr(), self.workspace.numel(), self.scratch_entries,
            self.status.data_ptr(), torch.cuda.current_stream(self.graph.device).cuda_stream)
        _cabi.check(rc, "gcc_sample_batch")
        self._last_steps = 1
        return q, k

    def sample_multi(self, first_sample_id: int, num_steps: int, stride: int | None = None, prof=None):
        """The batches of ``num_steps`` consecutive steps in one launch set (gcc_sample_multi): step t covers the sample
        ids ``first_sample_id + t * stride + [0, B)`` (``stride`` defaults to the batch size).  -> [(q, k)] per step, each
        pair in its own ring slot; every subgraph is bit for bit what :meth:`sample` gives for the same id."""
        import torch

        B = self.batch_size
        stride = B if stride is None else int(stride)
        pairs = []
        at = 0
        while at < num_steps:                      # more steps than one call may cover: several calls
            n = min(self.max_steps, num_steps - at)
            outs = (_cabi.GccBatchOut * (2 * n))()
            for t in range(n):
                views = self._ring[self._next]
                self._next = (self._next + 1) % len(self._ring)
                q, k = BatchedCSR(B, **views[0]), BatchedCSR(B, **views[1])
                outs[2 * t], outs[2 * t + 1] = q.c_struct(), k.c_struct()
                pairs.append((q, k))
            params = _cabi.GccSampleParams(
                run_seed=self.run_seed, first_sample_id=int(first_sample_id) + at * stride, batch_size=B,
                restart_u32=self.graph.restart_u32, seeds=None,
                prof=prof.handle if (prof is not None and at == 0) else None, hub_degree=self.hub_degree, max_hubs=self.max_hubs)
            rc = self.lib.gcc_sample_multi(
                self.graph.byref(), ctypes.byref(params), n, stride, outs, self.workspace.data_ptr(),
                self.workspace.numel(), self.scratch_entries, self.status.data_ptr(),
                torch.cuda.current_stream(self.graph.device).cuda_stream)
            _cabi.check(rc, "gcc_sample_multi")
            self._last_steps = n
            at += n
        return pairs

    _BITS = ((1, "induction scratch"), (2, "node capacity"), (4, "edge capacity"))

    def check_status(self) -> None:
        """Synchronising check of the device overflow flags (raises, never truncates).  The producer pipeline does not
        wait for this: it takes a :meth:`status_snapshot` per chunk and re-samples an overflowed chunk after
        :meth:`grow` (gcc_amd/train_step.py: BatchProducer)."""
        s = int(self.status.item())
        if s:
            what = [n for b, n in self._BITS if s & b]
            raise RuntimeError("gcc_sample_batch overflow: " + ", ".join(what) +
                               " -- construct DeviceRWRSampler with larger edge_cap/scratch_entries")

    # ---- overflow -> regrow -> re-sample (instead of killing a multi-hour run at the next log line)
    def status_snapshot(self):
        """Enqueue, on the current stream, a copy of the status word into pinned host memory followed by its reset:
        the word the token reads back covers exactly the calls issued since the previous snapshot.  -> token.
        Every snapshot in flight owns its own pinned word (a free list that grows on demand: a producer with many chunks
        in flight, or repeated re-issues, can never have a word overwritten before it was read); :meth:`read_snapshot`
        hands the word back."""
        import torch

        if self._snap is None:
            self._snap, self._snap_free = [], []
        if not self._snap_free:
            block = torch.zeros(16, dtype=torch.int32).pin_memory()
            base = len(self._snap)
            self._snap.extend(block[i:i + 1] for i in range(16))
            self._snap_free.extend(range(base + 15, base - 1, -1))
        i = self._snap_free.pop()
        self._snap[i].copy_(self.status, non_blocking=True)
        self.status.zero_()
        return i

    def snapshot_sync(self) -> None:
        """Wait for snapshots enqueued on the current stream (callers without an event of their own)."""
        import torch

        torch.cuda.current_stream(self.graph.device).synchronize()

    def read_snapshot(self, token) -> int:
        """The status bits of a snapshot whose stream work has completed (the caller synchronised on an event recorded
        after :meth:`status_snapshot`); the token's word returns to the free list (read a token once)."""
        bits = int(self._snap[token])
        self._snap_free.append(token)
        return bits

    def grow(self, bits: int, factor: int = 4, limit_bytes: int = 64 << 30) -> None:
        """Enlarge what overflowed: the induction scratch (bit 1; the workspace is re-allocated) and / or the edge
        capacity of every ring slot (bit 4).  Node capacity (bit 2) is an exact bound, B * (lmax + 1): its overflow is a
        sizing error and raises.  Replaced buffers are kept alive (launches in flight elsewhere may still use them)."""
        import torch

        if bits & 2:
            raise RuntimeError("gcc_sample_batch overflow: node capacity (node_cap must be batch_size * (lmax + 1))")
        if bits & 1:
            new = max(self.scratch_entries_per_step * factor, self._default_scratch)   # an undersized argument jumps to the heuristic first
            if new * self.max_steps * 4 > limit_bytes:
                raise RuntimeError(f"gcc_sample_batch overflow: induction scratch of {new * self.max_steps * 4 >> 20} MiB refused")
            self.scratch_entries_per_step = new
            self.scratch_entries = new * self.max_steps
            self._retired.append(self.workspace)
            self._alloc_workspace()
        if bits & 4:
            new = self.edge_cap * factor
            if new * 4 * 2 * len(self._ring) > limit_bytes:
                raise RuntimeError(f"gcc_sample_batch overflow: edge capacity of {new} per view refused")
            self.edge_cap = new
            for views in self._ring:
                for v in views:
                    self._retired.append(v["col_idx"])
                    v["col_idx"] = torch.zeros(new, dtype=torch.int32, device=self.graph.device)
        self.regrown += 1

    def last_seeds(self):
        """int32 device view of the seeds drawn by the most recent call ([B], or [steps * B] after sample_multi)."""
        import torch

        return self.workspace[: 4 * self.batch_size * getattr(self, "_last_steps", 1)].view(torch.int32)


# ------------------------------------------------------------------ reference API
def worker_init_fn(worker_id):
    """graph_dataset.py:23-30 loads a graph shard per DataLoader worker.  The
    device sampler runs in-process (HIP contexts do not survive fork), so there
    is nothing to initialise; kept so train.py's DataLoader call still type-checks."""
    return None


def batcher():
    """data_util.py:26-32.  Samples arrive pre-batched from the device, so the
    collate function only unwraps DataLoader's one-element list."""

    def batcher_dev(batch):
        if isinstance(batch, (list, tuple)) and len(batch) == 1:
            return batch[0]
        return batch

    return batcher_dev


class LoadBalanceGraphDataset:
    """Same constructor arguments and attributes (``total``, ``jobs``,
    ``num_samples``) as graph_dataset.py:33-80; iteration yields already-batched
    ``(graph_q, graph_k)`` of ``batch_size`` samples each.

    ``dgl_graphs_file`` is read without DGL (gcc_amd/ingest.py: read_dgl_graphs);
    ``graph`` overrides it: a :class:`DeviceGraph`, or ``(row_ptr, col_idx)``
    arrays, or a list of such pairs -- a multi-graph corpus: laid out in HBM shard
    by shard (``jobs``), batch i draws its seeds from worker shard i % num_workers
    as the reference's IterableDataset workers do.
    """

    def __init__(self, rw_hops=64, restart_prob=0.8, positional_embedding_size=32,
                 step_dist=[1.0, 0.0, 0.0], num_workers=1, dgl_graphs_file="./data/small.bin",
                 num_samples=10000, num_copies=1, graph_transform=None, aug="rwr", num_neighbors=5,
                 graph=None, batch_size=32, run_seed=0, device="cuda"):
        self.rw_hops = rw_hops
        self.num_neighbors = num_neighbors
        self.restart_prob = restart_prob
        self.positional_embedding_size = positional_embedding_size
        self.step_dist = step_dist
        self.num_samples = num_samples
        assert sum(step_dist) == 1.0
        assert positional_embedding_size > 1
        if list(step_dist) != [1.0, 0.0, 0.0]:
            raise NotImplementedError("only step_dist=[1,0,0] (train.py's only setting) is supported")
        assert aug in ("rwr", "ns")
        if aug != "rwr":
            raise NotImplementedError("aug='ns' is never selected by train.py (graph_dataset.py:131-162)")
        if graph_transform is not None:
            raise NotImplementedError("graph_transform is never set by train.py")
        self.aug = aug
        self.dgl_graphs_file = dgl_graphs_file
        self.graph_transform = graph_transform
        sizes = None
        if graph is None:
            graph, sizes = _load_graph_file(dgl_graphs_file)
        graphs = graph if isinstance(graph, list) else [graph]
        if not isinstance(graphs[0], DeviceGraph) and sizes is None:
            sizes = [len(rp) - 1 for rp, _ in graphs]
        if isinstance(graphs[0], DeviceGraph):
            assert len(graphs) == 1
            sizes = [graphs[0].num_nodes]
        # greedy LPT load balance of graph_dataset.py:63-77: graph indices per worker
        assert num_workers % num_copies == 0
        bins = max(num_workers // num_copies, 1)
        jobs = [list() for _ in range(bins)]
        workloads = [0] * bins
        for idx, size in sorted(enumerate(sizes), key=lambda t: t[1], reverse=True):
            argmin = workloads.index(min(workloads))
            workloads[argmin] += size
            jobs[argmin].append(idx)
        self.jobs = jobs * num_copies
        if isinstance(graphs[0], DeviceGraph):
            self.graph = graphs[0]                           # one graph: every worker holds it, one shard
            self.graph_order = [0]
        else:
            # Worker w samples among the nodes of ITS graphs jobs[w], concatenated in that order (worker_init_fn,
            # graph_dataset.py:23-30; __iter__ :85-92), and an IterableDataset worker yields whole batches: batch i is
            # worker i % num_workers's.  The corpus is laid out shard by shard in HBM, each shard with its own seed cdf;
            # gcc_sample_batch picks the shard from the batch index.  Workers without graphs (more workers than graphs)
            # would crash the reference (np.random.choice over nothing); they are left out of the rotation here.
            live = [j for j in jobs if j]
            self.graph_order = [i for j in live for i in j]
            rp, ci = _disjoint_union([graphs[i] for i in self.graph_order])
            shard_off = np.cumsum([0] + [sum(len(graphs[i][0]) - 1 for i in j) for j in live])
            self.graph = DeviceGraph(rp, ci, rw_hops=rw_hops, restart_prob=restart_prob, device=device,
                                     shard_off=shard_off if len(live) > 1 else None)
        self.total = self.num_samples * num_workers
        self.batch_size = batch_size
        self.run_seed = run_seed
        self._sampler = None          # built on first use: the MoCo path samples through its producer lanes instead
        self._epoch = 0

    @property
    def sampler(self):
        if self._sampler is None:
            self._sampler = DeviceRWRSampler(self.graph, self.batch_size, run_seed=self.run_seed)
        return self._sampler

    @property
    def node_cap(self):
        """Node capacity of one batch view (batch_size * (longest trace + 1)); known without allocating a sampler."""
        return self.batch_size * (self.graph.lmax + 1)

    def __len__(self):
        return self.total

    def __iter__(self):
        n_batch = self.total // self.batch_size
        # sample ids of an epoch start at a multiple of the batch size -- the kernel takes the worker shard of a batch from
        # sample_id // batch_size, and these are the ids the fused trainers use ((epoch - 1) * n_batch * batch_size): with
        # epoch * total a batch of epoch 2 on would straddle two shards whenever total % batch_size != 0
        base = self._epoch * n_batch * self.batch_size
        self._epoch += 1
        for i in range(n_batch):
            yield self.sampler.sample(base + i * self.batch_size)


def _disjoint_union(graphs):
    rps, cis, off, eoff = [np.zeros(1, dtype=np.int64)], [], 0, 0
    for rp, ci in graphs:
        rp = np.asarray(rp, dtype=np.int64)
        rps.append(rp[1:] + eoff)
        cis.append(np.asarray(ci, dtype=np.int64) + off)
        off += len(rp) - 1
        eoff += int(rp[-1])
    return np.concatenate(rps).astype(np.int32), np.concatenate(cis).astype(np.int32)


def _load_graph_file(path):
    """``dgl_graphs_file``: a DGL graph file as written by x2dgl.py (graph_dataset.py:26-29,58-60; container restated in
    gcc_amd/ingest.py, format DGL-recalled) or an .npz with row_ptr/col_idx.  -> (graphs, graph sizes for the LPT split)."""
    if str(path).endswith(".npz"):
        z = np.load(path)
        return [(z["row_ptr"], z["col_idx"])], None
    from .ingest import read_dgl_graphs

    graphs, labels = read_dgl_graphs(str(path))
    sizes = labels["graph_sizes"].tolist() if "graph_sizes" in labels else None      # graph_dataset.py:58-60
    return graphs, sizes
