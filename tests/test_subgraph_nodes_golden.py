"""Which nodes a random-walk trace selects, and in which order: oracle/sampler.py (Python and C) against node lists produced
by EXECUTING the reference's own ``_rwr_trace_to_dgl_graph`` (data_util.py:218-239; tests/golden/make_subgraph_golden.py ->
tests/golden/subgraph_nodes_reference.json): ``torch.unique`` of the trace ascending, the seed removed and put first, the seed
flag on local node 0.  (What DGL's ``subgraph`` does with that list stays DGL-recalled.)"""
import json
import os

import numpy as np

from gcc_amd.graphgen import powerlaw_graph
from oracle import sampler as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subgraph_nodes_reference.json")


def test_node_list_and_seed_position_are_the_reference_ones():
    gold = json.load(open(GOLD))
    assert gold["graph"] == "powerlaw_graph(3000, 30000, 3)"
    rp, ci = powerlaw_graph(3000, 30000, 3)
    assert len(gold["items"]) >= 10
    for it in gold["items"]:
        nodes, lrp, lci = O.py_subgraph(rp, ci, it["seed"], it["trace"])
        assert nodes == it["nodes"], it["seed"]
        assert it["seed_flag_at"] == [0] and nodes[0] == it["seed"]
        # induced CSR over that list: every parent edge between two members, and nothing else
        members = set(nodes)
        want = sum(1 for v in nodes for e in range(rp[v], rp[v + 1]) if int(ci[e]) in members)
        assert lrp[-1] == want == len(lci)
        assert all(0 <= c < len(nodes) for c in lci)
