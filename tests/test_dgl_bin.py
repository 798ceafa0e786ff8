"""DGL graph-file ingestion without DGL (graph_dataset.py:26-29,58-60, x2dgl.py:119-131) and the
LoadBalanceGraphDataset host logic (graph_dataset.py:34-80).  Format status: DGL-recalled, parity
unpinned (no DGL, no .bin in /root/reference) -- these tests pin the reader against the container description in
gcc_amd/ingest.py through a byte-level hand-built file and a writer round trip."""
import struct

import numpy as np
import pytest

from gcc_amd import ingest
from gcc_amd.graphgen import powerlaw_graph, tiny_graphs


def _tensor(a):
    code = {"i": 0, "u": 1, "f": 2}[a.dtype.kind]
    return (struct.pack("<QQiiiBBH", 0xDD5E40F096B4A13F, 0, 1, 0, a.ndim, code, a.dtype.itemsize * 8, 1)
            + struct.pack(f"<{a.ndim}q", *a.shape) + struct.pack("<q", a.nbytes) + a.tobytes())


def test_reader_on_a_file_assembled_byte_by_byte(tmp_path):
    """One 3-node path (0-1-2) written field by field as the header comment of gcc_amd/ingest.py describes."""
    indptr = np.array([0, 1, 3, 4], dtype=np.int64)
    indices = np.array([1, 0, 2, 1], dtype=np.int64)
    graph = _tensor(indptr) + _tensor(indices) + _tensor(np.arange(4, dtype=np.int64)) + struct.pack("<Q", 0) + struct.pack("<Q", 0)
    label = struct.pack("<Q", 1) + struct.pack("<Q", 11) + b"graph_sizes" + _tensor(np.array([3], dtype=np.int64))
    table_len = 8 + 3 * (8 + 8) + len(label)
    head = struct.pack("<QQQ", 0xDD2E4FF046B4A13F, 1, 1).ljust(4096, b"\0")
    table = (struct.pack("<Q", 1) + struct.pack("<QQ", 1, 4096 + table_len) + struct.pack("<QQ", 1, 3) + struct.pack("<QQ", 1, 4) + label)
    assert len(table) == table_len
    f = tmp_path / "one.bin"
    f.write_bytes(head + table + graph)
    graphs, labels = ingest.read_dgl_graphs(str(f))
    assert labels["graph_sizes"].tolist() == [3] and ingest.read_dgl_labels(str(f))["graph_sizes"].tolist() == [3]
    rp, ci = graphs[0]
    assert rp.dtype == np.int32 and rp.tolist() == [0, 1, 3, 4] and ci.tolist() == [1, 0, 2, 1]


def test_writer_reader_round_trip_and_idx_list(tmp_path):
    gs = [powerlaw_graph(3000, 30000, 1), powerlaw_graph(500, 3000, 2), tiny_graphs()["k4"], powerlaw_graph(1200, 9000, 3)]
    sizes = np.array([len(rp) - 1 for rp, _ in gs], dtype=np.int64)
    f = tmp_path / "small.bin"
    ingest.write_dgl_graphs(str(f), gs, labels={"graph_sizes": sizes})
    graphs, labels = ingest.read_dgl_graphs(str(f))
    assert labels["graph_sizes"].tolist() == sizes.tolist() and len(graphs) == 4
    for (rp, ci), (rp0, ci0) in zip(graphs, gs):
        assert np.array_equal(rp, rp0) and np.array_equal(ci, ci0)
    some, _ = ingest.read_dgl_graphs(str(f), idx_list=[3, 1])      # load_graphs(file, jobs[worker_id])
    assert np.array_equal(some[0][1], gs[3][1]) and np.array_equal(some[1][0], gs[1][0])


def test_reader_refuses_what_it_does_not_understand(tmp_path):
    f = tmp_path / "bad.bin"
    f.write_bytes(b"\0" * 5000)
    with pytest.raises(ValueError, match="not a DGL graph file"):
        ingest.read_dgl_graphs(str(f))
    f.write_bytes(struct.pack("<QQQ", 0xDD2E4FF046B4A13F, 2, 1).ljust(5000, b"\0"))
    with pytest.raises(ValueError, match="version 2"):
        ingest.read_dgl_graphs(str(f))
    ingest.write_dgl_graphs(str(f), [tiny_graphs()["path5"]], labels={"graph_sizes": np.array([5])})
    raw = bytearray(f.read_bytes())
    f.write_bytes(bytes(raw[:-10]))                                 # truncated
    with pytest.raises(ValueError):
        ingest.read_dgl_graphs(str(f))
    # a directed (non-symmetric) graph violates the sampler contract of x2dgl.py:39-62
    ingest.write_dgl_graphs(str(f), [(np.array([0, 1, 1]), np.array([1]))])
    with pytest.raises(ValueError):
        ingest.read_dgl_graphs(str(f))


class _FakeSampler:
    def __init__(self, graph, batch_size, run_seed=0):
        self.calls = []

    def sample(self, first_id):
        self.calls.append(first_id)
        return ("q%d" % first_id, "k%d" % first_id)


def test_load_balance_dataset_attributes_and_iteration(tmp_path, monkeypatch):
    """graph_dataset.py:34-80: LPT jobs, total = num_samples * num_workers, epoch offsets; data_util.py:26-32 batcher."""
    import gcc_amd.sampler as S

    class FakeGraph:                                                # no device in the CPU tier
        def __init__(self, rp, ci, **kw):
            self.num_nodes, self.lmax = len(rp) - 1, 40
            self.rp, self.ci, self.kw = rp, ci, kw

    monkeypatch.setattr(S, "DeviceGraph", FakeGraph)
    monkeypatch.setattr(S, "DeviceRWRSampler", _FakeSampler)
    gs = [powerlaw_graph(900, 5000, 1), powerlaw_graph(300, 2000, 2), powerlaw_graph(500, 3000, 3), tiny_graphs()["k4"]]
    sizes = np.array([len(rp) - 1 for rp, _ in gs], dtype=np.int64)
    f = tmp_path / "small.bin"
    ingest.write_dgl_graphs(str(f), gs, labels={"graph_sizes": sizes})
    ds = S.LoadBalanceGraphDataset(rw_hops=16, num_workers=4, num_copies=2, num_samples=40, dgl_graphs_file=str(f),
                                   batch_size=8)
    # two bins (num_workers // num_copies), greedy longest-processing-time: sizes sorted decreasing go to the lighter bin
    order = np.argsort(-sizes, kind="stable")
    bins, load = [[], []], [0, 0]
    for i in order:
        b = load.index(min(load))
        bins[b].append(int(i))
        load[b] += int(sizes[i])
    assert ds.jobs == bins * 2 and ds.total == 160 and len(ds) == 160 and ds.num_samples == 40
    assert ds.graph.num_nodes == int(sizes.sum())                  # disjoint union of all graphs
    # ... laid out shard by shard in jobs order, each worker shard a node range with its own seed cdf
    assert ds.graph_order == bins[0] + bins[1]
    assert ds.graph.kw["shard_off"].tolist() == [0, int(sizes[bins[0]].sum()), int(sizes.sum())]
    from tests.shard_check import reference_layout
    _, rp_ref, ci_ref, so_ref = reference_layout(gs, num_workers=4, num_copies=2)
    assert np.array_equal(ds.graph.rp, rp_ref) and np.array_equal(ds.graph.ci, ci_ref) and so_ref.tolist() == ds.graph.kw["shard_off"].tolist()
    assert ds.node_cap == 8 * 41
    first = list(ds)
    assert len(first) == 20 and ds.sampler.calls == [i * 8 for i in range(20)]
    second = list(ds)                                               # next epoch: fresh sample ids
    assert ds.sampler.calls[20:] == [160 + i * 8 for i in range(20)] and second[0] == ("q160", "k160")
    collate = S.batcher()
    assert collate([first[3]]) == first[3] and collate(first[3]) == first[3]
    assert S.worker_init_fn(0) is None
    with pytest.raises(NotImplementedError):
        S.LoadBalanceGraphDataset(dgl_graphs_file=str(f), aug="ns")


def test_rows_unsorted_far_into_the_file_are_sorted_and_label_count_is_checked(tmp_path):
    """DGL does not promise sorted rows: a file whose only unsorted rows come after the first thousand is sorted (every
    row is looked at), not refused; a graph_sizes label of the wrong length is refused."""
    rp, ci = powerlaw_graph(4000, 30000, 4)
    shuffled = ci.copy()
    rng = np.random.default_rng(0)
    assert len(rp) - 1 > 3000
    for v in range(2500, len(rp) - 1):                           # only late rows lose their order
        seg = shuffled[rp[v]:rp[v + 1]]
        if len(seg) > 1:
            seg[:] = seg[::-1] if np.all(np.diff(seg) > 0) else rng.permutation(seg)
    assert not np.array_equal(shuffled, ci)
    f = tmp_path / "late.bin"
    ingest.write_dgl_graphs(str(f), [(rp, shuffled)], labels={"graph_sizes": np.array([len(rp) - 1])})
    graphs, _ = ingest.read_dgl_graphs(str(f))
    assert np.array_equal(graphs[0][0], rp) and np.array_equal(graphs[0][1], ci)
    ingest.write_dgl_graphs(str(f), [(rp, ci)], labels={"graph_sizes": np.array([len(rp) - 1, 7])})
    with pytest.raises(ValueError, match="graph_sizes"):
        ingest.read_dgl_graphs(str(f))
