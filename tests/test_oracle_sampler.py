"""Pins the sampler oracle (oracle/sampler_oracle.c): Random123 Philox KATs,
the independent pure-Python restatement, committed golden vectors and
hand-derivable facts on tiny graphs.  CPU only."""
import json
import os

import numpy as np
import pytest

from gcc_amd.graph import max_nodes_per_seed_table, restart_threshold, seed_cdf_table
from gcc_amd.graphgen import powerlaw_graph, tiny_graphs
from oracle import sampler as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampler_golden.json")))


def _graph(name):
    g = GOLD["graphs"][name]
    return np.array(g["row_ptr"], np.int32), np.array(g["col_idx"], np.int32)


def test_philox_known_answers(coracle):
    for kat in GOLD["philox_kat"]:
        assert coracle.philox(kat["ctr"], kat["key"]).tolist() == kat["out"]
        assert O.py_philox4x32_10(kat["ctr"], kat["key"]) == kat["out"]


def test_host_tables_match_product():
    rp, _ = powerlaw_graph(2000, 20000, 4)
    assert np.array_equal(O.seed_cdf(rp), seed_cdf_table(rp))
    assert np.array_equal(O.max_nodes_table(700, 256, 0.8), max_nodes_per_seed_table(700, 256, 0.8))
    assert O.restart_threshold(0.8) == restart_threshold(0.8) == 3435973836
    # graph_dataset.py:113-124: exceeds rw_hops=256 only for in-degree > ~655
    t = O.max_nodes_table(1000, 256, 0.8)
    assert t[655] == 256 and t[700] > 256 and t[0] == 256


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"{c['graph']}-s{c['seed']}-g{c['g']}")
def test_c_oracle_reproduces_golden(coracle, case):
    rp, ci = _graph(case["graph"])
    thr = O.restart_threshold(case["restart_prob"])
    trace = coracle.rwr_trace(rp, ci, case["seed"], case["L"], case["run_seed"], case["g"], thr)
    assert trace.tolist() == case["trace"]
    r = coracle.sample_batch(rp, ci, [case["seed"]], [case["L"]], case["g"] & 1, case["run_seed"],
                             case["g"] >> 1, thr)
    assert r["parent_nid"].tolist() == case["nodes"]
    assert r["row_ptr"].tolist() == case["sub_row_ptr"]
    assert r["col_idx"].tolist() == case["sub_col"]


def test_seed_draw_golden(coracle):
    rp, _ = _graph("pl400")
    s = GOLD["pl400_seeds"]
    got = coracle.draw_seeds(O.seed_cdf(rp), s["run_seed"], s["first"], len(s["seeds"]))
    assert got.tolist() == s["seeds"]


def test_seed_distribution_follows_deg_075(coracle):
    rp, _ = powerlaw_graph(500, 5000, 2)
    cdf = O.seed_cdf(rp)
    seeds = coracle.draw_seeds(cdf, 3, 0, 200000)
    freq = np.bincount(seeds, minlength=len(rp) - 1) / 200000.0
    p = np.diff(np.concatenate([[0.0], cdf]))
    assert np.abs(freq - p).max() < 0.004


def test_hand_derivable_facts(coracle):
    thr = O.restart_threshold(0.8)
    g = tiny_graphs()
    # star, seed = leaf 3: every walk's first step is the centre 0; a second step goes to a leaf
    rp, ci = g["star6"]
    tr = coracle.rwr_trace(rp, ci, 3, 40, 1, 0, thr)
    assert tr[0] == 0 and set(tr.tolist()) <= set(range(6)) and 0 in tr
    r = coracle.sample_batch(rp, ci, [3], [40], 0, 1, 0, thr)
    nodes = r["parent_nid"].tolist()
    assert nodes[0] == 3 and nodes[1:] == sorted(nodes[1:]) and 3 not in nodes[1:]
    # induced star: centre (local 1) sees every other member, each leaf sees only the centre
    n = len(nodes)
    rows = [r["col_idx"][r["row_ptr"][i]:r["row_ptr"][i + 1]].tolist() for i in range(n)]
    c = nodes.index(0)
    assert rows[c] == sorted(set(range(n)) - {c}, key=lambda l: nodes[l])
    assert all(rows[i] == [c] for i in range(n) if i != c)
    # restart_prob = 1 - 2^-32 (~always restart): every walk has length 1 => trace = neighbours of the seed
    rp, ci = g["k4"]
    tr = coracle.rwr_trace(rp, ci, 2, 30, 9, 4, 0xFFFFFFFF)
    assert set(tr.tolist()) <= {0, 1, 3}
    # path graph: a walk of t steps cannot leave the t-neighbourhood of the seed
    rp, ci = g["path5"]
    tr = coracle.rwr_trace(rp, ci, 0, 25, 2, 0, 0)   # restart_u32 = 0: never restart -> one walk of 25 steps
    pos = [0] + tr.tolist()
    assert all(abs(a - b) == 1 for a, b in zip(pos, pos[1:]))


def test_exactly_L_entries_and_symmetric_subgraph(coracle):
    rp, ci = powerlaw_graph(3000, 30000, 5)
    cdf = O.seed_cdf(rp)
    lt = O.max_nodes_table(int(np.diff(rp).max()), 64, 0.8)
    seeds = coracle.draw_seeds(cdf, 1, 0, 16)
    L = lt[np.diff(rp)[seeds]]
    r = coracle.sample_batch(rp, ci, seeds, L, 1, 1, 0, O.restart_threshold(0.8))
    assert r["steps"] == int(L.sum())
    import scipy.sparse as sp
    N = r["node_off"][-1]
    a = sp.csr_matrix((np.ones(len(r["col_idx"])), r["col_idx"], r["row_ptr"]), shape=(N, N))
    assert (a != a.T).nnz == 0
    assert np.array_equal(r["parent_nid"][r["node_off"][:-1]], seeds)
