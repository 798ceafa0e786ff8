"""gcc_amd/ingest.py:read_edgelist against outputs of the reference's own ``data_util.Edgelist`` and
``NodeClassificationDataset._create_dgl_graph`` (executed by tests/golden/make_edgelist_golden.py with a recording DGLGraph;
committed as tests/golden/edgelist_reference.json): node ids by first appearance, one-hot labels, and the multigraph the
reference walks on = our simple symmetric CSR with ``edge_multiplicity`` copies of every edge."""
import json
import os
from collections import Counter

import numpy as np

from gcc_amd import ingest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "edgelist_reference.json")


def test_reader_reproduces_the_reference_reader_and_its_dgl_multigraph(tmp_path):
    gold = json.load(open(GOLD))
    (tmp_path / "toy.edgelist").write_text(gold["edgelist"])
    (tmp_path / "toy.nodelabel").write_text(gold["nodelabel"])
    d = ingest.read_edgelist(str(tmp_path / "toy.edgelist"), str(tmp_path / "toy.nodelabel"))
    assert {str(k): int(v) for k, v in d["node2id"].items()} == gold["node2id"]
    assert np.array_equal(np.asarray(d["y"]).astype(np.int64), np.asarray(gold["y"]))
    rp, ci = np.asarray(d["row_ptr"]), np.asarray(d["col_idx"])
    n = gold["num_nodes"]
    assert len(rp) - 1 == n
    ours = Counter()
    for v in range(n):
        for u in ci[rp[v]:rp[v + 1]]:
            ours[(v, int(u))] += d["edge_multiplicity"]
    ref = Counter((int(s), int(t)) for s, t in gold["dgl_edges"])
    assert ours == ref                                                   # the same directed multigraph, edge for edge
    assert d["edge_multiplicity"] == 2
    # in/out-degree of the reference's DGL graph = edge_multiplicity x simple degree: what max_nodes_out_degree_table is given
    outdeg = Counter(s for s, _ in gold["dgl_edges"])
    assert all(outdeg[v] == 2 * (rp[v + 1] - rp[v]) for v in range(n))
