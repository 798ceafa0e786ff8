"""Kernel-logic parity on CPU: gcc_amd/csrc/sampler.hip compiled for the
lock-step wave64 emulator (tests/hipemu) vs the C oracle, bit-exact."""
import numpy as np
import pytest

from gcc_amd.graphgen import powerlaw_graph, tiny_graphs
from oracle import sampler as O
from tests.hipemu.emu_driver import EmuGraph, emu_sample_batch

KEYS = ("node_off", "parent_nid", "row_ptr", "col_idx")


def _compare(coracle, rp, ci, g, B, run_seed, first, seeds=None, **kw):
    res, status, used = emu_sample_batch(g, B, run_seed, first, seeds=seeds, **kw)
    assert status == 0
    oseeds = coracle.draw_seeds(O.seed_cdf(rp), run_seed, first, B) if seeds is None else np.asarray(seeds, np.int32)
    assert used.tolist() == oseeds.tolist()
    L = g.ltab[np.minimum(np.diff(rp)[oseeds], len(g.ltab) - 1)]
    for view in range(2):
        ref = coracle.sample_batch(rp, ci, oseeds, L, view, run_seed, first, g.restart_u32)
        for k in KEYS:
            assert np.array_equal(res[view][k], ref[k]), (view, k)
        assert np.array_equal(res[view]["edge_off"], ref["edge_off"])
        gid = np.repeat(np.arange(B), np.diff(ref["node_off"]))
        assert np.array_equal(res[view]["graph_id"], gid)
    return res


@pytest.mark.parametrize("B,rw_hops,run_seed,first", [(1, 16, 0, 0), (5, 64, 11, 100), (9, 256, 2**40 + 5, 2**33)])
def test_powerlaw_matches_oracle(coracle, B, rw_hops, run_seed, first):
    rp, ci = powerlaw_graph(3000, 30000, 3)
    _compare(coracle, rp, ci, EmuGraph(rp, ci, rw_hops=rw_hops), B, run_seed, first)


def test_hub_seeds_exceed_rw_hops(coracle):
    # degrees > 655 make max_nodes_per_seed > rw_hops (graph_dataset.py:113-124): bigger LDS / sort sizes
    rp, ci = powerlaw_graph(20000, 400000, 1)
    deg = np.diff(rp)
    hubs = np.argsort(deg)[-3:].astype(np.int32)
    g = EmuGraph(rp, ci, rw_hops=64)
    assert g.ltab[deg[hubs]].min() > 128        # several sort sizes above the rw_hops floor
    _compare(coracle, rp, ci, g, 3, 7, 0, seeds=hubs)


def _graph_with_super_hub(n=20000, e=150000, spokes=9000, seed=4):
    """A power-law graph + one node joined to ``spokes`` others: its trace budget (deg^0.75 * e / (e - 1) / restart) passes
    1024, the small walk class's capacity, so the subgraphs seeded there go through the big-class launch."""
    rp, ci = powerlaw_graph(n, e, seed)
    src = np.repeat(np.arange(len(rp) - 1, dtype=np.int64), np.diff(rp))
    dst = ci.astype(np.int64)
    rng = np.random.default_rng(seed)
    hub = 17
    others = rng.choice(np.setdiff1d(np.arange(len(rp) - 1), [hub]), spokes, replace=False)
    src = np.concatenate([src, np.full(spokes, hub), others])
    dst = np.concatenate([dst, others, np.full(spokes, hub)])
    key = np.unique(src * (len(rp) - 1) + dst)                      # symmetric, sorted rows, no duplicates
    src, dst = key // (len(rp) - 1), key % (len(rp) - 1)
    rp2 = np.zeros(len(rp), np.int32)
    np.cumsum(np.bincount(src, minlength=len(rp) - 1), out=rp2[1:])
    return rp2, dst.astype(np.int32), hub


@pytest.mark.parametrize("hub_degree", [0, 4, -1])
def test_long_traces_take_the_big_walk_class(coracle, hub_degree):
    rp, ci, hub = _graph_with_super_hub()
    g = EmuGraph(rp, ci, rw_hops=64)
    deg = np.diff(rp)
    assert g.ltab[deg[hub]] > 1024 and g.lmax > 1024
    nb = ci[rp[hub]:rp[hub] + 2]
    # the hub itself (big class), two of its neighbours and drawn seeds (small class) in ONE launch; both orders
    _compare(coracle, rp, ci, g, 3, 5, 0, seeds=np.array([hub, nb[0], nb[1]], np.int32), hub_degree=hub_degree)
    _compare(coracle, rp, ci, g, 3, 5, 0, seeds=np.array([nb[0], hub, hub], np.int32), hub_degree=hub_degree)
    _compare(coracle, rp, ci, g, 6, 9, 40, hub_degree=hub_degree)


@pytest.fixture
def tiny_grids():
    """Static grids of a few workgroups (gcc_sampler_debug_grids): every induce workgroup walks through several virtual
    workgroups and subgraphs -- skipping those of the other size class --, every big-walk workgroup through several list
    entries.  The bench graphs only reach that regime on the 10M / 200M graph."""
    from tests.hipemu.emu_driver import emu_lib

    lib = emu_lib()

    def set_grids(small, big, walk_big):
        lib.gcc_sampler_debug_grids(small, big, walk_big)

    yield set_grids
    lib.gcc_sampler_debug_grids(0, 0, 0)


@pytest.mark.parametrize("grids", [(1, 1, 1), (3, 2, 2), (5, 1, 3)])
@pytest.mark.parametrize("hub_degree", [0, 4, -1])
def test_workgroups_walk_through_both_size_classes(coracle, tiny_grids, grids, hub_degree):
    rp, ci, hub = _graph_with_super_hub()
    g = EmuGraph(rp, ci, rw_hops=64)
    nb = ci[rp[hub]:rp[hub] + 3]
    big_deg = np.argsort(np.diff(rp))[-4:-1].astype(np.int32)             # the next largest rows: big induce class, small walk class
    seeds = np.array([nb[0], hub, big_deg[0], nb[1], hub, big_deg[1], nb[2], big_deg[2], hub], np.int32)
    tiny_grids(*grids)
    res = _compare(coracle, rp, ci, g, len(seeds), 5, 0, seeds=seeds, hub_degree=hub_degree)
    sizes = np.diff(res[0]["node_off"])
    assert (sizes > 320).sum() >= 3 and (sizes <= 320).sum() >= 3           # both induce classes are populated
    _compare(coracle, rp, ci, g, 7, 9, 40, hub_degree=hub_degree)            # drawn seeds


def test_long_traces_in_multi_step_launches(coracle):
    from tests.hipemu.emu_driver import emu_sample_multi

    rp, ci, hub = _graph_with_super_hub()
    g = EmuGraph(rp, ci, rw_hops=64)
    B, S = 3, 3
    cdf = O.seed_cdf(rp)
    first = next(f for f in range(0, 100000, B * S) if hub in coracle.draw_seeds(cdf, 13, f, B * S).tolist())
    pairs, status, seeds = emu_sample_multi(g, B, 13, first, S, B)
    assert status == 0
    big = 0
    for t, (q, k) in enumerate(pairs):
        single, st, used = emu_sample_batch(g, B, 13, first + t * B)
        assert st == 0
        big += int((g.ltab[np.minimum(np.diff(rp)[used], len(g.ltab) - 1)] > 1024).sum())
        for got, ref in ((q, single[0]), (k, single[1])):
            for key in KEYS + ("edge_off", "graph_id"):
                assert np.array_equal(got[key], ref[key]), (t, key)
    assert big >= 1, "pick a first id whose draws include the super hub"      # (seeds ~ deg^0.75: the hub is drawn often)


@pytest.mark.parametrize("name", ["path5", "star6", "tri_tail", "k4"])
def test_tiny_graphs(coracle, name):
    rp, ci = tiny_graphs()[name]
    g = EmuGraph(rp, ci, rw_hops=12)
    _compare(coracle, rp, ci, g, 4, 5, 0)
    _compare(coracle, rp, ci, g, 2, 5, 0, seeds=[0, len(rp) - 2])


def test_restart_prob_extremes(coracle):
    rp, ci = powerlaw_graph(1000, 8000, 6)
    for prob in (0.05, 0.5, 0.999):
        g = EmuGraph(rp, ci, rw_hops=64, restart_prob=prob)
        _compare(coracle, rp, ci, g, 3, 1, 0)


def test_overflow_is_flagged_not_truncated():
    rp, ci = powerlaw_graph(3000, 30000, 3)
    g = EmuGraph(rp, ci, rw_hops=64)
    _, status, _ = emu_sample_batch(g, 4, 1, 0, scratch_entries=8)
    assert status & 1
    _, status, _ = emu_sample_batch(g, 4, 1, 0, edge_cap=4)
    assert status & 4
    _, status, _ = emu_sample_batch(g, 4, 1, 0, node_cap=4)
    assert status & 2


def test_overflow_leaves_a_valid_structure():
    """A consumer that has not looked at `status` yet must never be able to index out of bounds."""
    rp, ci = powerlaw_graph(3000, 30000, 3)
    g = EmuGraph(rp, ci, rw_hops=64)
    for kw in (dict(scratch_entries=300), dict(edge_cap=40)):
        res, status, _ = emu_sample_batch(g, 6, 1, 0, **kw)
        assert status != 0
        for view in res:
            N = int(view["node_off"][-1])
            rptr = view["row_ptr"]
            cap = kw.get("edge_cap", 10**9)
            assert np.all(np.diff(rptr) >= 0) and rptr[0] == 0 and rptr[-1] <= max(cap, int(view["edge_off"][-1]))
            e = min(int(rptr[-1]), cap)
            cols = view["col_idx"][:e]
            assert cols.size == 0 or (cols.min() >= 0 and cols.max() < N)


def _dense_graph(n, p, seed):
    import scipy.sparse as sp

    rng = np.random.RandomState(seed)
    up = np.triu(rng.rand(n, n) < p, 1)
    up[np.arange(n - 1), np.arange(1, n)] = True
    a = sp.csr_matrix((up | up.T).astype(np.int8))
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def test_dense_subgraphs_drain_the_candidate_queue(coracle):
    """Almost every neighbour is a member (hit rate ~100 % instead of ~1.5 %): the per-wave queue of Bloom survivors
    (256 entries) is drained several times per unit and every unit's scratch slot fills up."""
    rp, ci = _dense_graph(300, 0.6, 2)
    g = EmuGraph(rp, ci, rw_hops=64, ltab=np.full(int(np.diff(rp).max()) + 1, 900, dtype=np.int32))
    res = _compare(coracle, rp, ci, g, 2, 3, 0)
    assert np.diff(res[0]["node_off"]).min() > 200 and len(res[0]["col_idx"]) > 40000


def test_many_units_per_subgraph_and_odd_row_alignment(coracle):
    """> 256 units in one subgraph (pack_kernel walks them in chunks of 256, four parts) and rows that start at every
    alignment inside a 16-byte quad; num_edges is not a multiple of 4 (the array's last quad is partial)."""
    rp, ci = _dense_graph(701, 0.9, 6)
    assert len(ci) % 4 != 0                                # (seed 5 gives a multiple of 4)
    g = EmuGraph(rp, ci, rw_hops=64, ltab=np.full(int(np.diff(rp).max()) + 1, 2500, dtype=np.int32))
    res = _compare(coracle, rp, ci, g, 1, 9, 0)
    n = int(res[0]["node_off"][-1])
    quads = sum((int(rp[v + 1]) + 3) // 4 - int(rp[v]) // 4 for v in res[0]["parent_nid"])
    assert n > 600 and quads > 256 * 256


def test_workgroups_cross_subgraph_boundaries(coracle):
    """More virtual workgroups than induce workgroups (G * 8): each workgroup takes several consecutive ones and walks
    from one subgraph into the next (subgraphs of unequal size, so that the boundaries fall inside the chunks)."""
    rp, ci = _dense_graph(701, 0.9, 5)
    deg = np.diff(rp)
    ltab = np.where(np.arange(deg.max() + 1) % 2 == 0, 2500, 40).astype(np.int32)   # long and short walks by seed degree
    g = EmuGraph(rp, ci, rw_hops=64, ltab=ltab)
    res = _compare(coracle, rp, ci, g, 3, 9, 0, seeds=[3, 600, 44])
    quads = [sum((int(rp[v + 1]) + 3) // 4 - int(rp[v]) // 4 for v in res[view]["parent_nid"][a:b])
             for view in range(2) for a, b in zip(res[view]["node_off"][:-1], res[view]["node_off"][1:])]
    assert sum(-(-q // 2048) for q in quads) > 2 * 6 * 8 and len(set(quads)) == 6      # chunk >= 3


def test_last_quad_of_col_idx_is_partial(coracle):
    for name in ("star6", "tri_tail"):                    # E = 10: the last row ends inside the array's last quad
        rp, ci = tiny_graphs()[name]
        assert len(ci) % 4 == 2
        g = EmuGraph(rp, ci, rw_hops=12)
        _compare(coracle, rp, ci, g, 3, 8, 0, seeds=[len(rp) - 2, len(rp) - 2, 0])


def test_multi_graph_corpus_samples_per_worker_shard(coracle):
    """LoadBalanceGraphDataset on a multi-graph corpus (graph_dataset.py:23-30,63-92): worker w draws seeds ~ deg^0.75
    normalised over ITS graphs only, and a DataLoader batch comes from one worker -- batch i from shard i % workers.
    Device kernel (emulator build) vs C oracle vs the Python restatement, bit-exact, over a rotation of batches."""
    from tests.shard_check import corpus, oracle_batch, reference_layout

    graphs = corpus()
    jobs, rp, ci, shard_off = reference_layout(graphs, num_workers=2)
    assert jobs == [[1], [3, 0, 2]]                                     # LPT: the 6000-node graph alone
    g = EmuGraph(rp, ci, rw_hops=32, shard_off=shard_off)
    B = 6
    hit = set()
    for batch in range(4):
        first = (10 + batch) * B
        seeds, views = oracle_batch(coracle, rp, ci, shard_off, g.ltab, g.restart_u32, B, 3, first)
        res, status, used = emu_sample_batch(g, B, 3, first)
        assert status == 0 and used.tolist() == seeds.tolist()
        for v in range(2):
            for k in KEYS:
                assert np.array_equal(res[v][k], views[v][k]), (batch, v, k)
        hit.add(int(np.searchsorted(shard_off, seeds[0], side="right") - 1))
    assert hit == {0, 1}
    # the unsharded draw over the union is a different distribution (what round 2 did): not the same seeds
    plain = coracle.draw_seeds(O.seed_cdf(rp), 3, 10 * B, B)
    assert plain.tolist() != oracle_batch(coracle, rp, ci, shard_off, g.ltab, g.restart_u32, B, 3, 10 * B)[0].tolist()


def test_shard_seed_distribution_is_deg_075_within_the_shard(coracle):
    from tests.shard_check import corpus, reference_layout

    _, rp, ci, shard_off = reference_layout(corpus(), num_workers=2)
    cdf = O.seed_cdf(rp, shard_off)
    B = 4096
    for sh in range(2):
        seeds = coracle.draw_seeds(cdf, 9, sh * B, B, shard_off=shard_off, batch_size=B)
        a, b = int(shard_off[sh]), int(shard_off[sh + 1])
        w = np.diff(rp)[a:b].astype(np.float64) ** 0.75
        p = w / w.sum()
        # compare the mass of the 20 heaviest nodes of the shard with its expectation (binomial, 5 sigma)
        top = np.argsort(w)[-20:]
        got = np.isin(seeds - a, top).mean()
        exp = p[top].sum()
        assert abs(got - exp) < 5 * np.sqrt(exp * (1 - exp) / B), (sh, got, exp)


def test_multi_step_call_equals_the_single_step_calls(coracle):
    """gcc_sample_multi: the batches of S consecutive steps in one launch set (sample ids first + t * stride + [0, B)) are,
    bit for bit, what S gcc_sample_batch calls produce -- and what the oracle says -- including the per-segment batch
    offsets, on a sharded corpus (the shard follows the batch index of every step) and with a rank stride."""
    from tests.hipemu.emu_driver import emu_sample_multi
    from tests.shard_check import corpus, oracle_batch, reference_layout

    _, rp, ci, shard_off = reference_layout(corpus(), num_workers=2)
    g = EmuGraph(rp, ci, rw_hops=32, shard_off=shard_off)
    B, S, first, stride = 5, 3, 7 * 5, 2 * 5                       # rank 1 of 2: ids 35.., 45.., 55..
    pairs, status, seeds = emu_sample_multi(g, B, 11, first, S, stride)
    assert status == 0 and len(pairs) == S
    for t in range(S):
        f = first + t * stride
        single, st1, used = emu_sample_batch(g, B, 11, f)
        assert st1 == 0 and seeds[t * B:(t + 1) * B].tolist() == used.tolist()
        oseeds, views = oracle_batch(coracle, rp, ci, shard_off, g.ltab, g.restart_u32, B, 11, f)
        assert used.tolist() == oseeds.tolist()
        for v in range(2):
            for k in KEYS + ("edge_off", "graph_id"):
                assert np.array_equal(pairs[t][v][k], single[v][k]), (t, v, k)
            for k in KEYS:
                assert np.array_equal(pairs[t][v][k], views[v][k]), (t, v, k)
    # limits are refused, not truncated
    with pytest.raises((RuntimeError, AssertionError), match="num_steps"):
        emu_sample_multi(g, B, 11, first, 17, stride)


# ---- hub rows are not scanned (sampler.hip: walk kernel tail / hub_write_kernel): same result, bit for bit
@pytest.mark.parametrize("max_hubs", [0, 1, 3, 32, 64])
@pytest.mark.parametrize("hub_degree", [1, 2, 3, 8, 40, -1])
def test_unscanned_hub_rows_give_the_same_subgraphs(coracle, hub_degree, max_hubs):
    """Rows of at least ``hub_degree`` are skipped by the induction; their induced rows are the mirror images of the other
    rows' hits (the parent graph is symmetric) + one search per pair of hubs.  At most ``max_hubs`` per subgraph (0 = the
    default, 8): a subgraph with more rows over the threshold raises ITS threshold to the next power of two that leaves few
    enough.  With the threshold at 1 and 32 slots EVERY row of a small ego-net is a hub (nothing is scanned at all: hub
    pairs only, the one-lane searches), with few slots the 16-lane searches run; -1 scans everything (the rounds 1-3
    path); the default 256 leaves these small graphs unaffected.  Always the C oracle's batches: node lists, row order,
    the seed's position inside a row."""
    rp, ci = powerlaw_graph(3000, 30000, 3)
    g = EmuGraph(rp, ci, rw_hops=64)
    _compare(coracle, rp, ci, g, 7, 21, 500, hub_degree=hub_degree, max_hubs=max_hubs)
    # hub seeds with several hundred members (far more rows over the threshold than hub slots: the threshold adapts)
    rp2, ci2 = powerlaw_graph(20000, 400000, 1)
    hubs = np.argsort(np.diff(rp2))[-2:].astype(np.int32)
    _compare(coracle, rp2, ci2, EmuGraph(rp2, ci2, rw_hops=64), 2, 7, 0, seeds=hubs, hub_degree=hub_degree, max_hubs=max_hubs)


@pytest.mark.parametrize("name", ["path5", "star6", "tri_tail", "k4"])
@pytest.mark.parametrize("hub_degree", [1, 2, 3])
def test_unscanned_hub_rows_on_tiny_graphs(coracle, name, hub_degree):
    rp, ci = tiny_graphs()[name]
    g = EmuGraph(rp, ci, rw_hops=12)
    _compare(coracle, rp, ci, g, 4, 5, 0, hub_degree=hub_degree)
    _compare(coracle, rp, ci, g, 2, 5, 0, seeds=[0, len(rp) - 2], hub_degree=hub_degree)


def test_unscanned_hub_rows_in_multi_step_launches(coracle):
    from tests.hipemu.emu_driver import emu_sample_multi

    rp, ci = powerlaw_graph(3000, 30000, 5)
    g = EmuGraph(rp, ci, rw_hops=32)
    B, S = 4, 3
    pairs, status, _ = emu_sample_multi(g, B, 9, 100, S, B, hub_degree=4)
    assert status == 0
    for t, (q, k) in enumerate(pairs):
        single, st, _ = emu_sample_batch(g, B, 9, 100 + t * B, hub_degree=-1)
        assert st == 0
        for got, ref in ((q, single[0]), (k, single[1])):
            for key in KEYS + ("edge_off", "graph_id"):
                assert np.array_equal(got[key], ref[key]), (t, key)


def test_unchecked_parent_scans_every_row(coracle):
    """A graph whose contract nobody checked (gcc_graph.flags without GCC_GRAPH_CONTRACT_CHECKED: DeviceGraph(validate=False)
    without trusted=True, raw C-API callers) never takes the hub-row short cut, whatever hub_degree says: the mirror images
    are the induced rows on a SYMMETRIC parent only.  An asymmetric sorted-row parent therefore still gives the subgraph a
    row-by-row induction (DGL's VertexSubgraph, the C oracle) gives."""
    rp0, ci0 = powerlaw_graph(3000, 30000, 3)
    rng = np.random.default_rng(0)
    keep = rng.random(len(ci0)) < 0.8
    keep[rp0[:-1]] = True                                  # no dead ends: every row keeps its first edge
    ci = ci0[keep]
    rp = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), rp0[:-1]))]).astype(np.int32)
    import scipy.sparse as sp
    a = sp.csr_matrix((np.ones(len(ci), np.int8), ci, rp), shape=(len(rp) - 1,) * 2)
    assert (a != a.T).nnz > 0
    g = EmuGraph(rp, ci, rw_hops=64, contract_checked=False)
    _compare(coracle, rp, ci, g, 7, 21, 500, hub_degree=2, max_hubs=32)
    _compare(coracle, rp, ci, g, 7, 21, 500)


@pytest.mark.parametrize("hub_degree,max_hubs", [(8, 3), (8, 32), (40, 32), (16, 8)])
def test_hub_pairs_through_the_parent_table(coracle, hub_degree, max_hubs):
    """With gcc_graph.hub_index / hub_adj (the adjacency among the parent's rows of at least hub_table_degree entries, built at
    upload: gcc_amd.graph.hub_tables) an edge between two unscanned hub rows is one bit probe instead of a search in the
    shorter row.  Same subgraphs bit for bit; a call whose threshold is BELOW the table's falls back to the searches."""
    rp, ci = powerlaw_graph(20000, 400000, 1)
    hubs = np.argsort(np.diff(rp))[-4:].astype(np.int32)
    g = EmuGraph(rp, ci, rw_hops=64, hub_table_degree=8)
    assert g.c.num_hubs > 100 and g.c.hub_table_degree == 8
    _compare(coracle, rp, ci, g, 4, 7, 0, seeds=hubs, hub_degree=hub_degree, max_hubs=max_hubs)
    _compare(coracle, rp, ci, g, 5, 3, 77, hub_degree=hub_degree, max_hubs=max_hubs)
    _compare(coracle, rp, ci, g, 3, 3, 99, hub_degree=4, max_hubs=max_hubs)      # below the table's threshold: searches


def test_two_classes_when_the_big_tables_are_small(coracle):
    """Hub seeds on a graph whose longest trace is a few hundred members (G1's regime): subgraphs over the small class's 320
    members go to the big class with tables for that trace.  (Written while a third, middle induce class was tried in round 5 --
    measured slower on the 10M / 200M graph, not kept: profiles/r5_induce_three_classes.txt -- whose first device run lost exactly
    these subgraphs.)"""
    rp, ci = powerlaw_graph(20000, 400000, 1)
    hubs = np.argsort(np.diff(rp))[-3:].astype(np.int32)
    g = EmuGraph(rp, ci, rw_hops=700)
    assert 320 < g.lmax + 1 < 1536
    res = _compare(coracle, rp, ci, g, 3, 7, 0, seeds=hubs)
    assert (np.diff(res[0]["node_off"]) > 320).any()
