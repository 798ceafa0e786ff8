"""gcc_gin_eval_fused (gcc_amd/csrc/encoder_eval.hip): the eval-mode encoder of generate.py:33-53 as one launch, one
workgroup per subgraph -- against the 15-launch chain in eval mode (same kernels' arithmetic) and oracle/encoder.py in
eval(), on a batch with an LDS-resident subgraph, one larger than the LDS capacity (global gather path), a two-node one
and an empty padding graph; edge multiplicity 2 and a non-zero seed position (the graph-classification datasets);
``embed_views`` = (f(q) + f(k)) / 2.  Emulator tier; the device tier is tests/test_generate_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import encoder as E
from tests.hipemu.emu_encoder import CpuBatch, emu_engine, reference_encoder


def _random_subgraph(rng, n, extra):
    pairs = {(i, i + 1) for i in range(n - 1)}
    extra = min(extra, n * (n - 1) // 2 - (n - 1))
    while len(pairs) < n - 1 + extra:
        a, b = sorted(rng.randint(0, n, 2))
        if a != b:
            pairs.add((a, b))
    rows = [[] for _ in range(n)]
    for a, b in pairs:
        rows[a].append(b)
        rows[b].append(a)
    return [sorted(r) for r in rows]


def _batch(sizes, seed, hub=False):
    rng = np.random.RandomState(seed)
    node_off, row_ptr, col = [0], [0], []
    for n in sizes:
        if n > 0:
            rows = _random_subgraph(rng, n, 2 * n) if n > 2 else ([[1], [0]] if n == 2 else [[]])
            if hub and n > 40:                         # a hub row (long rows cross lane-group chunks of the gather)
                for u in range(2, n, 2):
                    if u not in rows[0]:
                        rows[0].append(u)
                        rows[u].append(0)
                rows = [sorted(r) for r in rows]
            for r in rows:
                col += [node_off[-1] + u for u in r]
                row_ptr.append(len(col))
        node_off.append(node_off[-1] + max(n, 0))
    N = node_off[-1]
    pos = torch.nn.functional.normalize(torch.randn(N, 32, generator=torch.Generator().manual_seed(seed)), dim=1)
    return CpuBatch(dict(node_off=torch.tensor(node_off), row_ptr=torch.tensor(row_ptr), col_idx=torch.tensor(col),
                         pos_undirected=pos))


def _models(seed):
    torch.manual_seed(seed)
    oracle = E.OracleGraphEncoder()
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    model = reference_encoder()
    model.load_state_dict(oracle.state_dict())
    model._engine = emu_engine()
    model.eval()
    oracle.eval()
    return model, oracle


def _oracle_args(g, mult=1):
    n = g.n
    rp, ci = g.row_ptr[: n + 1].long(), g.col_idx.long()
    if mult > 1:                                        # the reference's doubled DGL graph: every edge `mult` times
        ci = ci.repeat_interleave(mult)
        rp = rp * mult
    return g.node_off.long(), rp, ci, g.pos_undirected[:n]


@pytest.mark.parametrize("mult", [1, 2])
def test_fused_eval_equals_the_chain_and_the_oracle(mult):
    model, oracle = _models(3)
    g = _batch([70, 300, 2, 0, 129, 33], seed=1, hub=True)
    g.edge_multiplicity = mult
    with torch.no_grad():
        model.fused_eval = True
        f_fused, p_fused = model(g, return_all_outputs=True)
        model.fused_eval = False
        f_chain, p_chain = model(g, return_all_outputs=True)
        ref, p_ref = oracle(*_oracle_args(g, mult), return_all_outputs=True)
    torch.testing.assert_close(f_fused, f_chain, rtol=1e-5, atol=2e-6)
    for a, b in zip(p_fused, p_chain):                           # (the neighbour sums run in another order: absolute to the tensor's scale)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=max(1e-4, 2e-6 * float(b.abs().max())))
    torch.testing.assert_close(f_fused, ref, rtol=1e-4, atol=2e-5)        # (graph 3 is empty: score = the prediction biases)
    for a, b in zip(p_fused, p_ref):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("case", ["mixed", "tiny", "dense", "dense4"])
def test_packed_small_subgraphs(case):
    """The packed kernel for subgraphs of 1 .. 64 nodes (a workgroup takes all of them whose last node lies in one window of
    64 node ids): windows with one subgraph and with dozens (more than one pooling round of 32: runs of 1- and 2-node graphs),
    a big subgraph or empty padding graphs in between, the first subgraph of a pack starting before its window, complete
    64-node graphs (the most CSR entries a pack can hold), more windows than one pass of the grid is not needed here."""
    rng = np.random.RandomState(7)
    if case == "mixed":
        sizes = [int(x) for x in rng.randint(2, 65, 40)] + [0, 0, 200, 3, 64, 64, 1, 0, 70, 5] + [int(x) for x in rng.randint(2, 30, 30)]
    elif case == "tiny":
        sizes = [1] * 70 + [2] * 50 + [0] * 5 + [1, 2, 3] * 20 + [64, 1, 1, 63, 2]
    elif case == "dense":
        sizes = [64, 63, 64, 2, 64, 61]
    else:                                                        # four complete 64-node graphs: 16128 entries, more than a run's column ids
        sizes = [64, 64, 64, 64, 5]                              # hold -> the run falls apart into its members
    model, oracle = _models(11)
    g = _batch(sizes, seed=5)
    if case.startswith("dense"):                                 # complete graphs: n (n - 1) CSR entries each
        node_off, row_ptr, col = [0], [0], []
        for n in sizes:
            for i in range(n):
                col += [node_off[-1] + u for u in range(n) if u != i]
                row_ptr.append(len(col))
            node_off.append(node_off[-1] + n)
        g = CpuBatch(dict(node_off=torch.tensor(node_off), row_ptr=torch.tensor(row_ptr), col_idx=torch.tensor(col),
                          pos_undirected=g.pos_undirected[: node_off[-1]]))
    g.seed_local = torch.tensor([int(rng.randint(0, max(n, 1))) for n in sizes], dtype=torch.int32)
    with torch.no_grad():
        model.fused_eval = True
        f_fused, p_fused = model(g, return_all_outputs=True)
        model.fused_eval = False
        f_chain, p_chain = model(g, return_all_outputs=True)
        ref = oracle(*_oracle_args(g), seed_local=g.seed_local.long())
    torch.testing.assert_close(f_fused, f_chain, rtol=1e-5, atol=2e-6)
    for a, b in zip(p_fused, p_chain):                           # (the neighbour sums run in another order: absolute to the tensor's scale)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=max(1e-4, 2e-6 * float(b.abs().max())))
    torch.testing.assert_close(f_fused, ref, rtol=1e-4, atol=2e-5)


def size_class_batch():
    """One subgraph on each side of every dispatch boundary of gcc_gin_eval_fused: 64 / 65 nodes (member of a run | a workgroup of
    its own), 320 / 321 nodes (LDS-resident kernel | general kernel), a 400-node hub ego-net, a complete graph on 130 nodes whose
    16770 CSR entries exceed the LDS-resident kernel's column-id space (general kernel by edge count), runs of one to four small
    subgraphs broken by larger ones and by the groups of four, three passes of 128 rows, a last pass with one wave's worth of rows."""
    sizes = [320, 64, 321, 65, 400, 257, 130, 3, 40, 17, 64, 9, 70, 1, 0, 33, 64, 64, 5]
    g = _batch(sizes, seed=9, hub=True)
    node_off = g.node_off.tolist()
    rp, ci = g.row_ptr[: node_off[-1] + 1].tolist(), g.col_idx.tolist()
    n0, n1, n = node_off[6], node_off[7], sizes[6]                # the 130-node subgraph becomes complete
    head_ci, tail_ci = ci[: rp[n0]], ci[rp[n1]:]
    mid_rp, mid_ci = [], []
    for i in range(n):
        mid_ci += [n0 + u for u in range(n) if u != i]
        mid_rp.append(len(head_ci) + len(mid_ci))                 # end of row n0 + i
    shift = mid_rp[-1] - rp[n1]
    rp = rp[: n0 + 1] + mid_rp + [x + shift for x in rp[n1 + 1:]]
    ci = head_ci + mid_ci + tail_ci
    g = CpuBatch(dict(node_off=torch.tensor(node_off), row_ptr=torch.tensor(rp), col_idx=torch.tensor(ci),
                      pos_undirected=g.pos_undirected[: node_off[-1]]))
    rng = np.random.RandomState(5)
    g.seed_local = torch.tensor([int(rng.randint(0, max(s, 1))) for s in sizes], dtype=torch.int32)
    return g


def test_three_size_classes():
    """size_class_batch through the fused call, the eval chain and the oracle."""
    model, oracle = _models(13)
    g = size_class_batch()
    with torch.no_grad():
        model.fused_eval = True
        f_fused, p_fused = model(g, return_all_outputs=True)
        model.fused_eval = False
        f_chain, p_chain = model(g, return_all_outputs=True)
        ref = oracle(*_oracle_args(g), seed_local=g.seed_local.long())
    torch.testing.assert_close(f_fused, f_chain, rtol=1e-5, atol=2e-6)
    for a, b in zip(p_fused, p_chain):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=max(1e-4, 2e-6 * float(b.abs().max())))
    torch.testing.assert_close(f_fused, ref, rtol=1e-4, atol=2e-5)


def test_multi_edges_beyond_the_small_kernels_edge_space():
    """A 64-node subgraph with every entry of the complete graph listed twice (8064 CSR entries: no simple graph, but the C ABI
    takes any CSR): too many column ids for the small kernel's LDS space, so the general kernel takes it; next to a simple one."""
    model, oracle = _models(17)
    node_off, row_ptr, col = [0, 64, 64 + 20], [0], []
    for i in range(64):
        col += [u for u in range(64) if u != i for _ in range(2)]
        row_ptr.append(len(col))
    for r in _random_subgraph(np.random.RandomState(3), 20, 30):
        col += [64 + u for u in r]
        row_ptr.append(len(col))
    pos = torch.nn.functional.normalize(torch.randn(84, 32, generator=torch.Generator().manual_seed(2)), dim=1)
    g = CpuBatch(dict(node_off=torch.tensor(node_off), row_ptr=torch.tensor(row_ptr), col_idx=torch.tensor(col), pos_undirected=pos))
    with torch.no_grad():
        model.fused_eval = True
        f_fused = model(g)
        model.fused_eval = False
        f_chain = model(g)
        ref = oracle(*_oracle_args(g))
    torch.testing.assert_close(f_fused, f_chain, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(f_fused, ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("sizes", [[20], [0], [65], [64, 65], [0, 0, 0], [321], [5, 0, 0, 0, 7], [1, 1, 0, 1]])
def test_tiny_batches(sizes):
    """Batches smaller than a group of four, made of empty graphs only (no edges: col_idx is NULL), of isolated nodes, one
    subgraph on either side of a size limit: the work list, the run finder's clamps at the batch's end."""
    model, oracle = _models(3)
    g = _batch(sizes, seed=1)
    with torch.no_grad():
        model.fused_eval = True
        f = model(g).clone()
        ref = oracle(*_oracle_args(g))
    torch.testing.assert_close(f, ref, rtol=1e-4, atol=2e-5)


def test_seed_position_and_view_mean():
    model, oracle = _models(4)
    q, k = _batch([40, 90, 17], seed=2), _batch([35, 61, 260], seed=3)
    q.seed_local = torch.tensor([5, 0, 16], dtype=torch.int32)        # entire_graph=True: data_util.py:236-237
    with torch.no_grad():
        emb = model.embed_views(q, k)
        fq = oracle(*_oracle_args(q), seed_local=q.seed_local.long())
        fk = oracle(*_oracle_args(k))
    assert tuple(emb.shape) == (3, 64)
    torch.testing.assert_close(emb, (fq + fk) / 2, rtol=1e-4, atol=2e-5)
    with torch.no_grad():                                              # entire_graph: both views are ONE graph object
        same = model.embed_views(q, q)
    torch.testing.assert_close(same, fq, rtol=1e-4, atol=2e-5)


def test_training_mode_passes_are_refused():
    model, _ = _models(5)
    model.train()
    g = _batch([20], seed=4)
    eng = model.engine()
    p, buf = eng.make_pass(model, g, training=True)
    with pytest.raises(RuntimeError, match="eval-mode"):
        eng.eval_fused([p])
