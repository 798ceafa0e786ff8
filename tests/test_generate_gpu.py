"""generate.py path on a real MI355X: NodeClassificationDataset + positional embedding + eval-mode encoder through the
C ABI, against the C sampler oracle (bit-exact node sets) and the torch encoder oracle run on the doubled multigraph
with the device's own positional embeddings."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_generate_path_against_oracles():
    from gcc_amd.datasets import NodeClassificationDataset
    from gcc_amd.encoder import GraphEncoder
    from gcc_amd.generate import test_moco as run_test_moco
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.posemb import DevicePosEmb
    from oracle import encoder as E
    from oracle import sampler as O

    rp, ci = powerlaw_graph(700, 4000, 3)
    n, B, mult = len(rp) - 1, 64, 2
    ds = NodeClassificationDataset("toy", rw_hops=48, restart_prob=0.8, positional_embedding_size=32, graph=(rp, ci),
                                   edge_multiplicity=mult, batch_size=B, run_seed=5)
    torch.manual_seed(0)
    oracle = E.OracleGraphEncoder()
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
    model = GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                         freq_embedding_size=16, degree_embedding_size=16, output_dim=64, node_hidden_dim=64,
                         edge_hidden_dim=64, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                         gnn_model="gin", degree_input=True).cuda()
    model.load_state_dict(oracle.state_dict())
    pe = DevicePosEmb(B, ds.sampler.node_cap, 32, device="cuda", seed=1, max_views=2, num_buffers=2)
    kept = []

    class Spy:
        def __iter__(self):
            for q, k in ds:
                torch.cuda.synchronize()
                yield q, k
                torch.cuda.synchronize()
                kept.append([(g.csr_numpy(), g.pos_undirected[: g.number_of_nodes()].cpu().clone(), g.valid) for g in (q, k)])

    emb = run_test_moco(Spy(), model, pe)
    ds.sampler.check_status()
    pe.check_status()
    assert emb.shape == (n, 64)
    c = O.COracle()
    deg = np.diff(rp)
    oracle.eval()
    ref = []
    for i, pair in enumerate(kept):
        seeds = np.zeros(B, dtype=np.int32)
        valid = pair[0][2]
        seeds[:valid] = np.arange(i * B, i * B + valid)
        L = ds.ltab[deg[seeds]]
        fs = []
        for view, (csr, pos, _) in enumerate(pair):
            r = c.sample_batch(rp, ci, seeds, L, view, 5, i * B, O.restart_threshold(0.8))
            assert (csr["parent_nid"] == r["parent_nid"]).all() and (csr["col_idx"] == r["col_idx"]).all()
            fs.append(oracle(torch.from_numpy(csr["node_off"].astype(np.int64)),
                             mult * torch.from_numpy(csr["row_ptr"].astype(np.int64)),
                             torch.repeat_interleave(torch.from_numpy(csr["col_idx"].astype(np.int64)), mult), pos).detach())
        ref.append(((fs[0] + fs[1]) / 2)[:valid])
    torch.testing.assert_close(emb, torch.cat(ref), rtol=1e-3, atol=1e-4)


def test_graph_classification_dataset_on_device():
    from gcc_amd import ingest
    from gcc_amd.datasets import GraphClassificationDataset
    from gcc_amd.encoder import GraphEncoder
    from gcc_amd.generate import test_moco as run_test_moco
    from gcc_amd.posemb import DevicePosEmb
    from oracle import encoder as E

    rng = np.random.RandomState(7)
    graphs = []
    for n in (9, 24, 61, 15, 150, 33, 420):
        pairs = {(i, i + 1) for i in range(n - 1)}
        while len(pairs) < 2 * n:
            a, b = sorted(rng.randint(0, n, 2))
            if a != b:
                pairs.add((a, b))
        rp, ci, _ = ingest.csr_from_pairs(np.array(sorted(pairs)), n)
        graphs.append((rp, ci))
    ds = GraphClassificationDataset("toy", graphs=graphs, batch_size=4, device="cuda")
    torch.manual_seed(3)
    oracle = E.OracleGraphEncoder()
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
    model = GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                         freq_embedding_size=16, degree_embedding_size=16, output_dim=64, node_hidden_dim=64,
                         edge_hidden_dim=64, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                         gnn_model="gin", degree_input=True).cuda()
    model.load_state_dict(oracle.state_dict())
    pe = DevicePosEmb(4, ds.node_cap, 32, device="cuda", seed=1, max_views=2, num_buffers=2)
    kept = []

    class Spy:
        def __iter__(self):
            for q, k in ds:
                yield q, k
                torch.cuda.synchronize()
                n = q.number_of_nodes()
                kept.append((q.node_off.cpu().long(), q.row_ptr[: n + 1].cpu().long(), q.col_idx.cpu().long(),
                             q.pos_undirected[:n].cpu().clone(), q.seed_local.cpu().long(), q.valid))

    emb = run_test_moco(Spy(), model, pe)
    pe.check_status()
    assert emb.shape == (7, 64)
    oracle.eval()
    ref = [oracle(no, rp, ci, pos, seed_local=sl).detach()[:valid] for no, rp, ci, pos, sl, valid in kept]
    torch.testing.assert_close(emb, torch.cat(ref), rtol=1e-3, atol=1e-4)


def test_fused_eval_kernel_equals_the_eval_chain_on_the_device():
    """gcc_gin_eval_fused (one launch, one workgroup per subgraph, generate.py:33-53) against gcc_gin_forward in eval mode
    on sampled batches with rw_hops 256 ego-nets (LDS-resident subgraphs and ones above the 256-row capacity) and against
    the torch oracle; edge multiplicity 2 as generate.py's datasets have it."""
    from gcc_amd.encoder import GraphEncoder
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from oracle import encoder as E
    from tests.headline_step_check import view_arrays

    rp, ci = powerlaw_graph(200_000, 2_000_000, 4)
    graph = DeviceGraph(rp, ci, rw_hops=256, device="cuda:0")
    B = 64
    smp = DeviceRWRSampler(graph, B, run_seed=2)
    pe = DevicePosEmb(B, smp.node_cap, 32, device="cuda:0", seed=2, max_views=2)
    q, k = smp.sample(0)
    pe.multi([q, k])
    smp.check_status()
    sizes = torch.diff(q.node_off[: B + 1]).cpu()
    assert int(sizes.max()) > 256 and int(sizes.min()) < 256            # both residency paths are exercised
    torch.manual_seed(3)
    oracle = E.OracleGraphEncoder()
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
    model = GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                         freq_embedding_size=16, degree_embedding_size=16, output_dim=64, node_hidden_dim=64,
                         edge_hidden_dim=64, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                         gnn_model="gin", degree_input=True).cuda()
    model.load_state_dict(oracle.state_dict())
    model.eval()
    oracle.eval()
    for mult in (1, 2):
        q.edge_multiplicity = k.edge_multiplicity = mult
        with torch.no_grad():
            model.fused_eval = True
            ff, pf = model(q, return_all_outputs=True)
            emb = model.embed_views(q, k)
            model.fused_eval = False
            fc, pc = model(q, return_all_outputs=True)
            fk = model(k)
        torch.testing.assert_close(ff, fc, rtol=1e-5, atol=5e-6)
        for a, b in zip(pf, pc):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=max(1e-3, 2e-6 * float(b.abs().max())))    # (neighbour sums in another order: absolute to the scale)
        torch.testing.assert_close(emb, (fc + fk) / 2, rtol=1e-5, atol=5e-6)
        (no, rpq, ciq), pos = view_arrays(q)
        with torch.no_grad():
            ref = oracle(no, mult * torch.from_numpy(rpq), torch.repeat_interleave(torch.from_numpy(ciq), mult), pos)
        torch.testing.assert_close(ff.cpu(), ref, rtol=1e-3, atol=1e-4)


def _device_batch(g):
    """tests/hipemu/emu_encoder.CpuBatch -> the same batch on the device"""
    for name in ("node_off", "row_ptr", "col_idx", "edge_off", "graph_id", "parent_nid", "pos_undirected", "seed_local"):
        if getattr(g, name, None) is not None:
            setattr(g, name, getattr(g, name).cuda())
    return g


@pytest.mark.parametrize("case", ["boundaries", "mixed", "tiny", "dense4"])
def test_fused_eval_size_classes_on_the_device(case):
    """The emulator tier's dispatch-boundary batches (tests/test_eval_fused_emu.py: one subgraph on each side of 64 / 65 and 320 /
    321 nodes, the edge-count limit, runs of one to four small subgraphs, empty graphs, a run whose entries do not fit) through
    gcc_gin_eval_fused on the GPU -- where wave_uniform() IS v_readfirstlane and a workgroup's LDS is real -- against the eval
    chain and the torch oracle."""
    import numpy as np

    from gcc_amd.encoder import GraphEncoder
    from oracle import encoder as E
    from tests import test_eval_fused_emu as T

    if case == "boundaries":
        g = T.size_class_batch()
    else:
        rng = np.random.RandomState(7)
        sizes = {"mixed": [int(x) for x in rng.randint(2, 65, 40)] + [0, 0, 200, 3, 64, 64, 1, 0, 70, 5] + [int(x) for x in rng.randint(2, 30, 30)],
                 "tiny": [1] * 70 + [2] * 50 + [0] * 5 + [1, 2, 3] * 20 + [64, 1, 1, 63, 2],
                 "dense4": [64, 64, 64, 64, 5]}[case]
        g = T._batch(sizes, seed=5)
        if case == "dense4":                                      # four complete graphs: more entries than a run's column ids hold
            node_off, row_ptr, col = [0], [0], []
            for n in sizes:
                for i in range(n):
                    col += [node_off[-1] + u for u in range(n) if u != i]
                    row_ptr.append(len(col))
                node_off.append(node_off[-1] + n)
            g = T.CpuBatch(dict(node_off=torch.tensor(node_off), row_ptr=torch.tensor(row_ptr), col_idx=torch.tensor(col),
                                pos_undirected=g.pos_undirected[: node_off[-1]]))
        g.seed_local = torch.tensor([int(rng.randint(0, max(n, 1))) for n in sizes], dtype=torch.int32)
    torch.manual_seed(11)
    oracle = E.OracleGraphEncoder()
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    oracle.eval()
    with torch.no_grad():
        ref = oracle(*T._oracle_args(g), seed_local=g.seed_local.long())
    model = GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                         freq_embedding_size=16, degree_embedding_size=16, output_dim=64, node_hidden_dim=64,
                         edge_hidden_dim=64, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                         gnn_model="gin", degree_input=True).cuda()
    model.load_state_dict(oracle.state_dict())
    model.eval()
    g = _device_batch(g)
    with torch.no_grad():
        model.fused_eval = True
        ff, pf = model(g, return_all_outputs=True)
        ff, pf = ff.clone(), [x.clone() for x in pf]
        model.fused_eval = False
        fc, pc = model(g, return_all_outputs=True)
    torch.testing.assert_close(ff, fc, rtol=1e-5, atol=5e-6)
    for a, b in zip(pf, pc):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=max(1e-3, 2e-6 * float(b.abs().max())))
    torch.testing.assert_close(ff.cpu(), ref, rtol=1e-3, atol=1e-4)
