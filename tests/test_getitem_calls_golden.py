"""What the reference asks of DGL's random walk for a seed (graph_dataset.py:94-130), recorded by EXECUTING the reference's own
``LoadBalanceGraphDataset.__getitem__`` with ``random_walk_with_restart`` replaced by a recorder
(tests/golden/make_getitem_golden.py -> tests/golden/getitem_calls_reference.json): both views start at the sampled node
(step_dist [1, 0, 0]), the restart probability is passed through, and the node budget is
max(rw_hops, int(in_degree ** 0.75 * e / (e - 1) / restart_prob + 0.5)) -- the table ``gcc_amd.graph`` uploads as ``ltab``."""
import json
import os

from oracle import sampler as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "getitem_calls_reference.json")


def test_node_budget_table_and_call_arguments_are_the_reference_ones():
    items = json.load(open(GOLD))
    assert len(items) >= 40
    from gcc_amd.graph import max_nodes_out_degree_table

    lb = [it for it in items if "family" not in it]
    gd = [it for it in items if it.get("family") == "GraphDataset"]
    assert len(lb) >= 40 and len(gd) >= 20
    for it in lb:                                                          # LoadBalanceGraphDataset: in-degree ** 0.75
        tab = O.max_nodes_table(600, it["rw_hops"], it["restart_prob"])
        assert int(tab[it["in_degree"]]) == it["max_nodes_per_seed"], it
        assert it["seeds"] == [it["in_degree"], it["in_degree"]]          # node id = its in-degree in the generator's parent graph
    assert {(it["rw_hops"], it["restart_prob"]) for it in lb} == {(256, 0.8), (64, 0.8), (16, 0.5), (4, 0.05)}
    for it in gd:                                                          # GraphDataset family (generate.py): out-degree, no power
        tab = max_nodes_out_degree_table(600, it["rw_hops"], it["restart_prob"])
        assert int(tab[it["out_degree"]]) == it["max_nodes_per_seed"], it
        assert it["seeds"] == [it["out_degree"], it["out_degree"]]


def test_device_table_is_the_oracle_table():
    import numpy as np

    from gcc_amd.graph import max_nodes_per_seed_table as dev_table

    for hops, rp in ((256, 0.8), (16, 0.5)):
        assert np.array_equal(np.asarray(dev_table(600, hops, rp)), O.max_nodes_table(600, hops, rp))


def test_graph_classification_seed_rule_and_whole_graph_views_are_the_reference_ones():
    """GraphClassificationDataset (graph_dataset.py:311-345): seed = out_degrees().argmax() (first maximum), both views are the
    whole graph in its own node order, the seed flag sits on the seed (data_util.py:225-237 with entire_graph=True)."""
    import numpy as np

    from gcc_amd.graph import max_nodes_out_degree_table

    items = [it for it in json.load(open(GOLD)) if it.get("family") == "GraphClassificationDataset"]
    assert len(items) == 3
    for it in items:
        deg = np.asarray(it["out_degrees"])
        seed = int(np.argmax(deg))                                       # what gcc_amd.datasets.GraphClassificationDataset._convert_idx does
        assert it["seeds"] == [seed, seed] and it["seed_flag_at"] == [seed]
        assert it["subgraph_nodes"] == list(range(len(deg)))
        assert int(max_nodes_out_degree_table(int(deg.max()), 8, it["restart_prob"])[deg[seed]]) == it["max_nodes_per_seed"]
