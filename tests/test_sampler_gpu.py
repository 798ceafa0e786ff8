"""Parity tests proper: the HIP sampler (through the C ABI and the Python host)
vs the CPU oracle, bit-exact, on a real MI355X."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("node_off", "parent_nid", "graph_id", "row_ptr", "col_idx")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "the -m gpu tier needs a GPU"
    return torch


def _oracle_views(coracle, rp, ci, seeds, L, run_seed, first, thr):
    out = []
    for view in range(2):
        r = coracle.sample_batch(rp, ci, seeds, L, view, run_seed, first, thr)
        r["graph_id"] = np.repeat(np.arange(len(seeds)), np.diff(r["node_off"])).astype(np.int32)
        out.append(r)
    return out


def _check(coracle, torch, rp, ci, B, rw_hops, run_seed, first, restart_prob=0.8, seeds=None, **sampler_kw):
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.sampler import DeviceRWRSampler
    from oracle import sampler as O

    g = DeviceGraph(rp, ci, rw_hops=rw_hops, restart_prob=restart_prob)
    s = DeviceRWRSampler(g, B, run_seed=run_seed, **sampler_kw)
    dseeds = None if seeds is None else torch.tensor(seeds, dtype=torch.int32, device="cuda")
    q, k = s.sample(first, seeds=dseeds)
    s.check_status()
    oseeds = coracle.draw_seeds(O.seed_cdf(rp), run_seed, first, B) if seeds is None else np.asarray(seeds, np.int32)
    assert s.last_seeds().cpu().numpy().tolist() == oseeds.tolist()
    lt = O.max_nodes_table(int(np.diff(rp).max()), rw_hops, restart_prob)
    L = lt[np.diff(rp)[oseeds]]
    ref = _oracle_views(coracle, rp, ci, oseeds, L, run_seed, first, O.restart_threshold(restart_prob))
    for view, gb in enumerate((q, k)):
        got = gb.csr_numpy()
        for key in KEYS:
            assert np.array_equal(got[key], ref[view][key]), (view, key)
        assert np.array_equal(got["edge_off"], ref[view]["edge_off"])
    return q, k, ref


@pytest.mark.parametrize("B,rw_hops,run_seed,first", [(1, 16, 0, 0), (7, 64, 11, 100), (32, 256, 2**40 + 5, 2**33),
                                                      (64, 256, 1, 12345)])
def test_small_powerlaw_bit_exact(coracle, torch_cuda, B, rw_hops, run_seed, first):
    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(20000, 200000, 3)
    _check(coracle, torch_cuda, rp, ci, B, rw_hops, run_seed, first)


@pytest.mark.parametrize("hub_degree,max_hubs", [(8, 1), (8, 3), (40, 32), (2, 32), (-1, 0)])
def test_unscanned_hub_rows_bit_exact(coracle, torch_cuda, hub_degree, max_hubs):
    """Thresholds far below the default make most rows of these small ego-nets hubs (mirror images + pair searches only); few
    slots raise the per-subgraph threshold; -1 scans everything.  Always the C oracle's batches."""
    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(3000, 30000, 3)
    _check(coracle, torch_cuda, rp, ci, 32, 64, 21, 500, hub_degree=hub_degree, max_hubs=max_hubs)


@pytest.mark.parametrize("name", ["path5", "star6", "tri_tail", "k4"])
def test_tiny_graphs(coracle, torch_cuda, name):
    from gcc_amd.graphgen import tiny_graphs

    rp, ci = tiny_graphs()[name]
    _check(coracle, torch_cuda, rp, ci, 4, 12, 5, 0)
    _check(coracle, torch_cuda, rp, ci, 2, 12, 5, 0, seeds=[0, len(rp) - 2])


def test_restart_prob_extremes(coracle, torch_cuda):
    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(5000, 40000, 6)
    for prob in (0.05, 0.5, 0.999):
        _check(coracle, torch_cuda, rp, ci, 8, 64, 1, 0, restart_prob=prob)


def test_committed_golden_vectors(torch_cuda):
    """tests/golden/sampler_golden.json was written by the pure-Python restatement."""
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.sampler import DeviceRWRSampler

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampler_golden.json")))
    cache = {}
    for case in gold["cases"]:
        key = (case["graph"], case["L"])
        if key not in cache:
            g = gold["graphs"][case["graph"]]
            dg = DeviceGraph(np.array(g["row_ptr"], np.int32), np.array(g["col_idx"], np.int32),
                             rw_hops=case["L"], restart_prob=case["restart_prob"])
            if dg.lmax != case["L"]:     # golden cases use L == rw_hops (low degrees)
                continue
            cache[key] = dg
        dg = cache[key]
        s = DeviceRWRSampler(dg, 1, run_seed=case["run_seed"])
        views = s.sample(case["g"] >> 1, seeds=torch_cuda.tensor([case["seed"]], dtype=torch_cuda.int32, device="cuda"))
        s.check_status()
        got = views[case["g"] & 1].csr_numpy()
        assert got["parent_nid"].tolist() == case["nodes"]
        assert got["row_ptr"].tolist() == case["sub_row_ptr"]
        assert got["col_idx"].tolist() == case["sub_col"]


def test_hub_seeds_large_L(coracle, torch_cuda):
    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(200000, 4000000, 1)
    hubs = np.argsort(np.diff(rp))[-8:].astype(np.int32)
    _check(coracle, torch_cuda, rp, ci, 8, 256, 7, 0, seeds=hubs.tolist())


@pytest.mark.parametrize("grids", [(2, 1, 1), (64, 8, 3)])
def test_size_classes_with_tiny_grids(coracle, torch_cuda, grids):
    """Both size classes of the walk and of the induction in one launch (hub seeds: long traces, hundreds of members; their
    neighbours: the rw_hops budget), with the static grids capped (gcc_sampler_debug_grids) so that every workgroup walks
    through several virtual workgroups / subgraphs / list entries -- the regime the 10M / 200M graph runs in."""
    from gcc_amd import _cabi
    from tests.test_sampler_emu import _graph_with_super_hub

    rp, ci, top = _graph_with_super_hub(200000, 4000000, 30000, 1)        # its trace budget is ~4500 entries: the big walk class
    order = np.argsort(np.diff(rp))
    assert order[-1] == top
    hubs = order[-12:].astype(np.int32)
    small = np.array([ci[rp[h]] for h in hubs], np.int32)                # a neighbour of every hub
    seeds = np.stack([hubs, small], 1).reshape(-1)
    lib = _cabi.load()
    lib.gcc_sampler_debug_grids(*grids)
    try:
        for hd in (0, -1):
            _check(coracle, torch_cuda, rp, ci, len(seeds), 256, 7, 0, seeds=seeds.tolist(), hub_degree=hd,
                   scratch_entries=1 << 26, edge_cap=1 << 24)           # (a hub-only batch: beyond the sizing heuristics)
        _check(coracle, torch_cuda, rp, ci, 64, 256, 3, 640)
    finally:
        lib.gcc_sampler_debug_grids(0, 0, 0)


def test_full_size_g1_bit_exact_and_properties(coracle, torch_cuda):
    """BASELINE config 2's graph (1M nodes / 10M edges), bsz 256, rw_hops 256."""
    import scipy.sparse as sp

    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(1_000_000, 10_000_000, 0)
    for first in (0, 256 * 977):
        q, k, ref = _check(coracle, torch_cuda, rp, ci, 256, 256, 0, first)
        for gb in (q, k):
            c = gb.csr_numpy()
            N = c["node_off"][-1]
            a = sp.csr_matrix((np.ones(len(c["col_idx"]), np.int8), c["col_idx"], c["row_ptr"]), shape=(N, N))
            assert (a != a.T).nnz == 0                       # induced subgraph of a symmetric graph
            assert a.diagonal().sum() == 0
            blocks = np.repeat(np.arange(256), np.diff(c["node_off"]))
            assert np.array_equal(blocks[a.tocoo().row], blocks[a.tocoo().col])   # block diagonal (dgl.batch)
            for b in range(0, 256, 37):                      # seed first, rest sorted ascending
                seg = c["parent_nid"][c["node_off"][b]:c["node_off"][b + 1]]
                assert np.all(np.diff(seg[1:]) > 0) and seg[0] not in seg[1:]


def test_config4_like_dense_graph_bit_exact(coracle, torch_cuda):
    """BASELINE configs[3]'s regime at a fifth of its size (2M nodes / 40M edges, average degree 40): a full batch drawn
    as the product draws it, and a batch of hub seeds -- several consecutive virtual workgroups per induce workgroup,
    subgraphs with > 1000 members and hundreds of units each."""
    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(2_000_000, 40_000_000, 0)
    _check(coracle, torch_cuda, rp, ci, 256, 256, 3, 256 * 41)
    hubs = np.argsort(np.diff(rp))[-512::8].astype(np.int32)     # 64 of the 512 largest degrees
    # (a batch of nothing but hubs is 3x what the default scratch is sized for: the status word says so, see test_overflow_flag)
    q, k, ref = _check(coracle, torch_cuda, rp, ci, 64, 256, 9, 0, seeds=hubs.tolist(), scratch_entries=160 << 20,
                       edge_cap=64 << 20)
    units = [int(((rp[m + 1] + 3) // 4 - rp[m] // 4).sum() + 255) // 256
             for r in ref for m in (r["parent_nid"][a:b] for a, b in zip(r["node_off"][:-1], r["node_off"][1:]))]
    assert sum((u + 15) // 16 for u in units) > 2 * 128 * 8      # > 2 virtual workgroups per induce workgroup
    assert max(np.diff(ref[0]["node_off"])) > 1000


def test_config4_full_size_bit_exact(coracle, torch_cuda, g2_graph):
    """BASELINE configs[3] at its REAL size (10M nodes / 200M edges requested, rw_hops 256, restart 0.8, bsz 256): one batch
    drawn as the product draws it and one batch of hub seeds, node ids and batched CSR of both views bit for bit the C
    oracle's (graph_dataset.py:94-130, data_util.py:218-239); then the multi-step launch bench.py --mode sampler times
    (16 steps per call) against single-step calls of the same sample ids."""
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.sampler import DeviceRWRSampler

    rp, ci = g2_graph
    assert len(rp) - 1 > 9_900_000 and len(ci) > 199_000_000
    _check(coracle, torch_cuda, rp, ci, 256, 256, 0, 256 * 5)
    hubs = np.argsort(np.diff(rp))[-512::8].astype(np.int32)     # 64 of the 512 largest degrees (up to 12,649 neighbours)
    # (a batch of nothing but hubs is far beyond the sizing heuristics: the scratch is sized from the oracle's own batch --
    #  one 1024-entry slot per unit of 256 aligned quads of the members' parent rows, every row counted)
    from oracle import sampler as O
    lt = O.max_nodes_table(int(np.diff(rp).max()), 256, 0.8)
    views = _oracle_views(coracle, rp, ci, hubs, lt[np.diff(rp)[hubs]], 9, 0, O.restart_threshold(0.8))
    units = sum(int(((rp[m + 1] + 3) // 4 - rp[m] // 4).sum() + 255) // 256 + 1
                for r in views for m in (r["parent_nid"][a:b] for a, b in zip(r["node_off"][:-1], r["node_off"][1:])))
    q, k, ref = _check(coracle, torch_cuda, rp, ci, 64, 256, 9, 0, seeds=hubs.tolist(), scratch_entries=int(1.05 * 1024 * units),
                       edge_cap=max(int(1.05 * max(len(r["col_idx"]) for r in views)), 1 << 20))
    assert max(np.diff(ref[0]["node_off"])) > 1000
    # the launch shape of the published sampler line: 16 steps per call == 16 single-step calls
    g = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, validate=False, trusted=True)
    multi = DeviceRWRSampler(g, 256, run_seed=0, num_buffers=16, max_steps=16)
    single = DeviceRWRSampler(g, 256, run_seed=0)
    pairs = multi.sample_multi(256 * 32, 16, 256)
    multi.check_status()
    for t in (0, 7, 15):
        a = single.sample(256 * (32 + t))
        single.check_status()
        for x, y in zip(a, pairs[t]):
            cx, cy = x.csr_numpy(), y.csr_numpy()
            for key in KEYS:
                assert np.array_equal(cx[key], cy[key]), (t, key)


def test_overflow_flag(torch_cuda):
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.sampler import DeviceRWRSampler

    rp, ci = powerlaw_graph(20000, 200000, 3)
    g = DeviceGraph(rp, ci, rw_hops=64)
    s = DeviceRWRSampler(g, 8, edge_cap=16)
    s.sample(0)
    with pytest.raises(RuntimeError, match="edge capacity"):
        s.check_status()


def test_multi_graph_corpus_samples_per_worker_shard(coracle):
    """SURVEY.md 8 a-1 / a-2 on the device: LoadBalanceGraphDataset over a multi-graph corpus draws every DataLoader
    batch's seeds from ONE worker shard (graph_dataset.py:23-30,63-92), bit-exact against the C oracle."""
    from gcc_amd.sampler import LoadBalanceGraphDataset
    from tests.shard_check import corpus, oracle_batch, reference_layout

    graphs = corpus()
    jobs, rp, ci, shard_off = reference_layout(graphs, num_workers=2)
    B = 16
    ds = LoadBalanceGraphDataset(rw_hops=32, num_workers=2, num_copies=1, num_samples=4 * B, graph=graphs, batch_size=B,
                                 run_seed=3, device="cuda:0")
    assert ds.jobs == jobs and ds.graph.num_shards == 2
    assert np.array_equal(ds.graph.row_ptr.cpu().numpy(), rp)
    shards = []
    for i, (q, k) in enumerate(ds):                       # total // B = 8 batches of epoch 0
        ds.sampler.check_status()
        seeds, views = oracle_batch(coracle, rp, ci, shard_off, ds.graph.ltab.cpu().numpy(), ds.graph.restart_u32, B, 3, i * B)
        assert ds.sampler.last_seeds().cpu().numpy().tolist() == seeds.tolist()
        for gb, ref in zip((q, k), views):
            got = gb.csr_numpy()
            for key in ("node_off", "parent_nid", "row_ptr", "col_idx"):
                assert np.array_equal(got[key], ref[key]), (i, key)
        shards.append(int(np.searchsorted(shard_off, seeds[0], side="right") - 1))
    assert shards == [0, 1] * 4
