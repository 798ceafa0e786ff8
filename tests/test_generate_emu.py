"""generate.py path: edge-list ingestion, NodeClassificationDataset on the sampler (one item per
node, same seed for both views, out-degree L rule, multigraph multiplicity) and test_moco -- kernels on the emulator,
checked against the C sampler oracle (bit-exact) and the torch encoder oracle (on the doubled multigraph)."""
import numpy as np
import pytest
import torch

from gcc_amd import ingest
from gcc_amd.datasets import NodeClassificationDataset
from gcc_amd.generate import test_moco as run_test_moco
from gcc_amd.graph import max_nodes_out_degree_table
from gcc_amd.posemb import DevicePosEmb
from oracle import encoder as E
from oracle import sampler as O
from tests.hipemu.emu_driver import EmuGraph, emu_lib, emu_sample_batch
from tests.hipemu.emu_encoder import CpuBatch, emu_engine, reference_encoder

HID = 32


def _write_edgelist(tmp_path, n=60, extra=90, seed=0, repeat=1):
    rng = np.random.RandomState(seed)
    ids = rng.permutation(1000)[:n] + 5                     # arbitrary node names: re-indexed by first appearance
    pairs = {(i, i + 1) for i in range(n - 1)}
    while len(pairs) < n - 1 + extra:
        a, b = sorted(rng.randint(0, n, 2))
        if a != b:
            pairs.add((a, b))
    pairs = sorted(pairs)
    lines = [f"{ids[a]} {ids[b]}" for a, b in pairs]
    if repeat == 2:
        lines += [f"{ids[b]} {ids[a]}" for a, b in pairs]     # a file that lists both directions
    (tmp_path / "toy.edgelist").write_text("\n".join(lines) + "\n")
    labels = rng.randint(0, 3, n)
    (tmp_path / "toy.nodelabel").write_text("\n".join(f"{ids[i]} {labels[i] + 7}" for i in range(n)) + "\n")
    return ids, pairs, labels


def test_edgelist_reader_follows_the_reference_reindexing(tmp_path):
    ids, pairs, labels = _write_edgelist(tmp_path)
    d = ingest.read_edgelist(str(tmp_path / "toy.edgelist"), str(tmp_path / "toy.nodelabel"))
    n = len(ids)
    # ids in order of first appearance (data_util.py:79-83)
    order = []
    for a, b in pairs:
        for x in (ids[a], ids[b]):
            if x not in order:
                order.append(x)
    assert [d["node2id"][x] for x in order] == list(range(n))
    rp, ci = d["row_ptr"], d["col_idx"]
    assert d["edge_multiplicity"] == 2                       # (x,y)+(y,x) in Edgelist, both again in _create_dgl_graph
    assert len(ci) == 2 * len(pairs) and (np.diff(rp) > 0).all()
    dense = np.zeros((n, n), int)
    for a, b in pairs:
        dense[d["node2id"][ids[a]], d["node2id"][ids[b]]] = dense[d["node2id"][ids[b]], d["node2id"][ids[a]]] = 1
    for v in range(n):
        assert list(ci[rp[v]:rp[v + 1]]) == list(np.nonzero(dense[v])[0])       # rows sorted, symmetric
    # labels: one-hot over label ids in order of first appearance (data_util.py:92-109)
    assert d["y"].shape[0] == n and (d["y"].sum(1) == 1).all()
    _, _, _ = _write_edgelist(tmp_path, repeat=2)
    assert ingest.read_edgelist(str(tmp_path / "toy.edgelist"))["edge_multiplicity"] == 4
    (tmp_path / "bad.edgelist").write_text("1 2\n2 3\n1 2\n")
    with pytest.raises(ValueError):
        ingest.read_edgelist(str(tmp_path / "bad.edgelist"))       # non-uniform multiplicity
    (tmp_path / "loop.edgelist").write_text("1 2\n2 2\n")
    with pytest.raises(ValueError):
        ingest.read_edgelist(str(tmp_path / "loop.edgelist"))


def _emu_dataset(rp, ci, mult, B, rw_hops=24):
    ltab = max_nodes_out_degree_table(int(np.diff(rp).max()), rw_hops, 0.8, mult)
    g = EmuGraph(rp, ci, rw_hops=rw_hops, restart_prob=0.8, ltab=ltab)
    calls = []

    def sample_fn(first_id, seeds):
        res, status, used = emu_sample_batch(g, B, 3, first_id, seeds=seeds)
        assert status == 0 and (used == seeds).all()
        calls.append((first_id, seeds.copy(), res))
        out = []
        for r in res:
            n = len(r["parent_nid"])
            b = CpuBatch(dict(node_off=torch.from_numpy(r["node_off"].astype(np.int64)),
                              row_ptr=torch.from_numpy(r["row_ptr"].astype(np.int64)),
                              col_idx=torch.from_numpy(r["col_idx"].astype(np.int64)),
                              pos_undirected=torch.zeros(n, HID)), node_cap=B * (g.lmax + 1))
            b.parent_nid[:n] = torch.from_numpy(r["parent_nid"])
            out.append(b)
        return tuple(out)

    ds = NodeClassificationDataset("toy", rw_hops=rw_hops, restart_prob=0.8, positional_embedding_size=HID,
                                   graph=(rp, ci), edge_multiplicity=mult, batch_size=B, sample_fn=sample_fn)
    return ds, g, calls


def test_dataset_walks_every_node_in_order_with_the_out_degree_rule(tmp_path):
    _write_edgelist(tmp_path)
    d = ingest.read_edgelist(str(tmp_path / "toy.edgelist"))
    rp, ci, mult = d["row_ptr"], d["col_idx"], d["edge_multiplicity"]
    B = 16
    ds, g, calls = _emu_dataset(rp, ci, mult, B)
    assert len(ds) == ds.total == len(rp) - 1
    batches = list(ds)
    assert len(batches) == ds.num_batches() == 4 and batches[-1][0].valid == 60 - 48
    c = O.COracle()
    deg = np.diff(rp)
    for i, (first, seeds, res) in enumerate(calls):
        assert first == i * B and list(seeds[: batches[i][0].valid]) == list(range(i * B, i * B + batches[i][0].valid))
        L = np.array([max(24, int(mult * deg[s] * np.e / (np.e - 1) / 0.8 + 0.5)) for s in seeds], dtype=np.int32)  # :244-255
        assert (g.ltab[deg[seeds]] == L).all()
        for view in range(2):                                # same seed for both views (other_node_idx = node_idx, :238)
            ref = c.sample_batch(rp, ci, seeds, L, view, 3, first, O.restart_threshold(0.8))
            assert (res[view]["parent_nid"] == ref["parent_nid"]).all()
            assert (res[view]["node_off"] == ref["node_off"]).all()
            assert (res[view]["col_idx"] == ref["col_idx"]).all()
            assert (res[view]["parent_nid"][res[view]["node_off"][:-1]] == seeds).all()     # seed first
    assert batches[0][0].edge_multiplicity == mult


def test_generate_embeddings_match_the_oracle_on_the_multigraph(tmp_path):
    _write_edgelist(tmp_path, n=40, extra=50, seed=2)
    d = ingest.read_edgelist(str(tmp_path / "toy.edgelist"))
    rp, ci, mult = d["row_ptr"], d["col_idx"], d["edge_multiplicity"]
    B = 16
    ds, g, calls = _emu_dataset(rp, ci, mult, B)
    torch.manual_seed(1)
    oracle = E.OracleGraphEncoder()
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
    model = reference_encoder()
    model.load_state_dict(oracle.state_dict())
    model._engine = emu_engine()
    pe = DevicePosEmb(B, B * (g.lmax + 1), HID, device="cpu", lib=emu_lib(), ptr=lambda t: 0 if t is None else t.data_ptr(),
                      max_views=2, num_buffers=4)
    kept = []

    class Spy:                                               # keeps the batches so the oracle can see the same inputs
        def __iter__(self):
            for q, k in ds:
                kept.append((q, k))
                yield q, k

    emb = run_test_moco(Spy(), model, pe)
    assert emb.shape == (40, 64) and not model.training
    oracle.eval()
    ref = []
    for q, k in kept:
        fs = []
        for b in (q, k):
            n = b.n
            fs.append(oracle(b.node_off.long(), mult * b.row_ptr[: n + 1].long(),
                             torch.repeat_interleave(b.col_idx.long(), mult), b.pos_undirected[:n]).detach())
        ref.append(((fs[0] + fs[1]) / 2)[: q.valid])
    torch.testing.assert_close(emb, torch.cat(ref), rtol=1e-4, atol=2e-5)


def test_graph_classification_dataset_whole_graphs_and_seed_flag():
    """GraphClassificationDataset (entire_graph=True): every item is a whole graph in its own node order, the seed flag
    sits on the max-out-degree node, both views are the same graph; embeddings against the oracle."""
    from gcc_amd.datasets import GraphClassificationDataset

    rng = np.random.RandomState(4)
    graphs = []
    for n in (7, 19, 33, 12, 70):
        pairs = {(i, i + 1) for i in range(n - 1)}
        while len(pairs) < 2 * n:
            a, b = sorted(rng.randint(0, n, 2))
            if a != b:
                pairs.add((a, b))
        rp, ci, m = ingest.csr_from_pairs(np.array(sorted(pairs)), n)
        assert m == 1
        graphs.append((rp, ci))
    ds = GraphClassificationDataset("toy", graphs=graphs, batch_size=4, device="cpu")
    assert len(ds) == ds.total == 5 and ds.entire_graph
    batches = list(ds)
    assert len(batches) == 2 and batches[0][0] is batches[0][1] and batches[1][0].valid == 1
    g0 = batches[0][0]
    assert g0.node_off.tolist() == [0, 7, 26, 59, 71]
    seeds = g0.ndata["seed"].numpy()
    for b, (rp, ci) in enumerate(graphs[:4]):
        lo = int(g0.node_off[b])
        assert np.nonzero(seeds[lo:lo + len(rp) - 1])[0].tolist() == [int(np.argmax(np.diff(rp)))]
        assert (g0.col_idx[int(g0.edge_off[b]):int(g0.edge_off[b + 1])].numpy() - lo == ci).all()      # own node order
    torch.manual_seed(2)
    oracle = E.OracleGraphEncoder()
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
    model = reference_encoder()
    model.load_state_dict(oracle.state_dict())
    model._engine = emu_engine()
    pe = DevicePosEmb(4, ds.node_cap, HID, device="cpu", lib=emu_lib(), ptr=lambda t: 0 if t is None else t.data_ptr(),
                      max_views=2, num_buffers=2)
    kept = []

    class Spy:
        def __iter__(self):
            for q, k in ds:
                kept.append(q)
                yield q, k

    emb = run_test_moco(Spy(), model, pe)
    assert emb.shape == (5, 64)
    oracle.eval()
    ref = []
    for g in kept:
        n = int(g.node_off[-1])
        f = oracle(g.node_off.long(), g.row_ptr[: n + 1].long(), g.col_idx.long(), g.pos_undirected[:n],
                   seed_local=g.seed_local.long()).detach()
        ref.append(f[: g.valid])
    torch.testing.assert_close(emb, torch.cat(ref), rtol=1e-4, atol=2e-5)


def test_tudataset_reader(tmp_path):
    """raw TU layout -> per-graph CSR in file node order, labels re-indexed in ascending order; malformed input refused"""
    from gcc_amd import ingest

    # graph 1: path 1-2-3 (nodes 1..3), graph 2: triangle 4-5-6 plus pendant 7, graph 3: single edge 8-9; labels -1, 1, -1
    und = [(1, 2), (2, 3), (4, 5), (5, 6), (4, 6), (6, 7), (8, 9)]
    lines = [f"{u}, {v}" for u, v in und] + [f"{v}, {u}" for u, v in und]
    rng = np.random.default_rng(0)
    rng.shuffle(lines)
    (tmp_path / "IMDB-BINARY_A.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "IMDB-BINARY_graph_indicator.txt").write_text("\n".join(map(str, [1, 1, 1, 2, 2, 2, 2, 3, 3])) + "\n")
    (tmp_path / "IMDB-BINARY_graph_labels.txt").write_text("-1\n1\n-1\n")
    d = ingest.read_tudataset(str(tmp_path), "imdb-binary")
    assert d["num_labels"] == 2 and d["graph_labels"].tolist() == [0, 1, 0]
    want = [([0, 1, 3, 4], [1, 0, 2, 1]), ([0, 2, 4, 7, 8], [1, 2, 0, 2, 0, 1, 3, 2]), ([0, 1, 2], [1, 0])]
    assert len(d["graphs"]) == 3
    for (rp, ci), (wrp, wci) in zip(d["graphs"], want):
        assert rp.tolist() == wrp and ci.tolist() == wci and rp.dtype == np.int32
    # malformed files are refused, not repaired
    (tmp_path / "IMDB-BINARY_A.txt").write_text("\n".join(lines[:-1]) + "\n")
    with pytest.raises(ValueError, match="symmetric"):
        ingest.read_tudataset(str(tmp_path), "IMDB-BINARY")
    (tmp_path / "IMDB-BINARY_A.txt").write_text("\n".join(lines + ["3, 4", "4, 3"]) + "\n")
    with pytest.raises(ValueError, match="different graphs"):
        ingest.read_tudataset(str(tmp_path), "imdb-binary")
    (tmp_path / "IMDB-BINARY_A.txt").write_text("\n".join(lines + ["1, 2"]) + "\n")
    with pytest.raises(ValueError, match="repeated"):
        ingest.read_tudataset(str(tmp_path), "imdb-binary")
