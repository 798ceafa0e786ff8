"""Kernel-logic parity on CPU for the GIN encoder: gcc_amd/csrc/encoder*.hip on
the wave64 emulator vs the reference-generated golden vectors and the oracle."""
import os

import torch

from oracle import encoder as E
from tests.hipemu.emu_encoder import CpuBatch, emu_engine, reference_encoder

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_golden.pt"), weights_only=False)
TOL = dict(rtol=1e-4, atol=2e-5)     # north_star: 1e-3 rel; the f32 path is far inside it


def _set_bn_train(model):
    model.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.train()


def test_state_dict_is_a_drop_in():
    enc = reference_encoder()
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == GOLD["state_dict_shapes"]
    assert [n for n, _ in enc.named_parameters()] == GOLD["param_names"]
    enc.load_state_dict(GOLD["moco"]["init"]["model"], strict=True)


def test_forward_q_and_k_match_reference_golden():
    g = GOLD["moco"]
    model, ema = reference_encoder(), reference_encoder()
    model.load_state_dict(g["init"]["model"])
    ema.load_state_dict(g["init"]["model_ema"])
    model.train()
    _set_bn_train(ema)
    eng = emu_engine()
    bq, bk = CpuBatch(GOLD["views"][0]), CpuBatch(GOLD["views"][1])
    pq, bufq = eng.make_pass(model, bq, training=True, keep=g["masks"].contiguous(), slot=0)
    pk, bufk = eng.make_pass(ema, bk, training=True, keep=None, slot=1)
    eng.forward([pq, pk])
    torch.testing.assert_close(bufq["feat"], g["feat_q"], **TOL)
    torch.testing.assert_close(bufk["feat"], g["feat_k"], **TOL)
    # SumPooling outputs (all_outputs of gin.py:232)
    for i, ref in enumerate(g["all_outputs_q"]):
        torch.testing.assert_close(bufq["pooled"][i + 1].float(), ref, rtol=1e-4, atol=1e-3)
    # BatchNorm running statistics after one training forward
    after = g["after"]["model"]
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            torch.testing.assert_close(v, after[k], rtol=1e-4, atol=1e-5, msg=k)


def test_eval_mode_matches_oracle():
    g = GOLD["moco"]
    model = reference_encoder()
    model.load_state_dict(g["after"]["model"])
    model.eval()
    eng = emu_engine()
    bq = CpuBatch(GOLD["views"][0])
    pq, bufq = eng.make_pass(model, bq, training=False)
    eng.forward([pq])
    torch.testing.assert_close(bufq["feat"], g["feat_eval"], **TOL)


def test_long_rows_and_ragged_tiles_match_oracle():
    """a star-like batch: the hub row's edges are split over seven lane groups of the edge-balanced gather (side slots,
    combined in group order), rows inside a group's chunk finish there; N is not a multiple of the tile."""
    import numpy as np

    torch.manual_seed(3)
    n0, n1 = 150, 7
    edges = [(0, i) for i in range(1, n0)] + [(i, i + 1) for i in range(1, n0 - 1)]
    edges += [(n0 + i, n0 + j) for i in range(n1) for j in range(i + 1, n1)]
    import scipy.sparse as sp
    e = np.array(edges)
    a = sp.csr_matrix((np.ones(2 * len(e)), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n0 + n1,) * 2)
    a.sort_indices()
    view = dict(node_off=torch.tensor([0, n0, n0 + n1]), row_ptr=torch.from_numpy(a.indptr.astype(np.int64)),
                col_idx=torch.from_numpy(a.indices.astype(np.int64)), pos_undirected=torch.randn(n0 + n1, 32))
    oracle = E.OracleGraphEncoder()
    model = reference_encoder()
    model.load_state_dict(oracle.state_dict())
    oracle.train()
    model.train()
    keep = (torch.rand(5, 2, 64) > 0.5).float()
    ref = oracle(view["node_off"], view["row_ptr"], view["col_idx"], view["pos_undirected"], dropout_masks=keep)
    eng = emu_engine()
    p, buf = eng.make_pass(model, CpuBatch(view), training=True, keep=keep)
    eng.forward([p])
    torch.testing.assert_close(buf["feat"], ref.detach(), **TOL)


def test_rows_without_edges_inside_a_tile_match_oracle():
    """nodes without induced edges between rows that have some (generate.py's whole-graph batches can hold them): the
    edge-balanced gather finds the row of an edge by searching the tile's row pointers, where such rows repeat a value."""
    import numpy as np
    import scipy.sparse as sp

    torch.manual_seed(5)
    n = 70                                                  # two tiles; rows 3, 4, 40 and the last one have no edges
    iso = {3, 4, 40, n - 1}
    keep = [i for i in range(n) if i not in iso]
    edges = [(keep[i], keep[i + 1]) for i in range(len(keep) - 1)] + [(keep[0], k) for k in keep[2:30]]
    e = np.array(edges)
    a = sp.csr_matrix((np.ones(2 * len(e)), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n, n))
    a.sort_indices()
    assert all(a.indptr[i] == a.indptr[i + 1] for i in iso)
    view = dict(node_off=torch.tensor([0, 45, n]), row_ptr=torch.from_numpy(a.indptr.astype(np.int64)),
                col_idx=torch.from_numpy(a.indices.astype(np.int64)), pos_undirected=torch.randn(n, 32))
    oracle = E.OracleGraphEncoder()
    model = reference_encoder()
    model.load_state_dict(oracle.state_dict())
    oracle.train()
    model.train()
    keep_mask = (torch.rand(5, 2, 64) > 0.5).float()
    ref = oracle(view["node_off"], view["row_ptr"], view["col_idx"], view["pos_undirected"], dropout_masks=keep_mask)
    eng = emu_engine()
    p_, buf = eng.make_pass(model, CpuBatch(view), training=True, keep=keep_mask)
    eng.forward([p_])
    torch.testing.assert_close(buf["feat"], ref.detach(), **TOL)


def test_edge_multiplicity_equals_the_doubled_multigraph():
    """NodeClassificationDataset (graph_dataset.py:296-304 on top of data_util.py:84-85) hands the encoder a graph in
    which every edge exists twice: in-degrees and neighbour sums double.  edge_multiplicity = 2 on the simple CSR
    must equal the oracle run on the CSR with every entry duplicated (eval mode, as generate.py:40)."""
    import numpy as np
    import scipy.sparse as sp

    torch.manual_seed(5)
    n0, n1 = 90, 40
    rng = np.random.RandomState(0)
    edges = [(0, i) for i in range(1, n0)] + [(int(rng.randint(1, n0)), int(rng.randint(1, n0))) for _ in range(150)]
    edges += [(n0 + i, n0 + (i + 1) % n1) for i in range(n1)]
    e = np.array([(i, j) for i, j in edges if i != j])
    a = sp.csr_matrix((np.ones(2 * len(e)), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n0 + n1,) * 2)
    a.sum_duplicates()
    a.data[:] = 1
    a.sort_indices()
    rp, ci = a.indptr.astype(np.int64), a.indices.astype(np.int64)
    pos = torch.randn(n0 + n1, 32)
    node_off = torch.tensor([0, n0, n0 + n1])
    oracle = E.OracleGraphEncoder()
    oracle.degree_embedding.weight.data.normal_()
    for m in oracle.modules():                       # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
    oracle.eval()
    ref = oracle(node_off, torch.from_numpy(2 * rp), torch.from_numpy(np.repeat(ci, 2)), pos)
    model = reference_encoder()
    model.load_state_dict(oracle.state_dict())
    model.eval()
    b = CpuBatch(dict(node_off=node_off, row_ptr=torch.from_numpy(rp), col_idx=torch.from_numpy(ci), pos_undirected=pos))
    b.edge_multiplicity = 2
    eng = emu_engine()
    p, buf = eng.make_pass(model, b, training=False)
    eng.forward([p])
    torch.testing.assert_close(buf["feat"], ref.detach(), **TOL)
    b.edge_multiplicity = 1
    p1, buf1 = eng.make_pass(model, b, training=False)
    eng.forward([p1])
    assert (buf1["feat"] - ref.detach()).abs().max() > 1e-2          # the multiplicity matters


def _grads_after_backward(model):
    return {n: p.grad for n, p in model.named_parameters() if p.grad is not None}


def test_backward_matches_reference_golden():
    g = GOLD["moco"]
    model = reference_encoder()
    model.load_state_dict(g["init"]["model"])
    model.train()
    eng = emu_engine()
    bq = CpuBatch(GOLD["views"][0])
    pq, bufq = eng.make_pass(model, bq, training=True, keep=g["masks"].contiguous())
    eng.forward([pq])
    eng.backward(model, pq, bufq, g["dfeat_q"].clone())
    got = _grads_after_backward(model)
    # every parameter the reference gives a gradient to gets one here, and nothing else is touched
    assert set(g["grads"]) <= set(got)
    for n, ref in g["grads"].items():
        scale = max(float(ref.abs().max()), 1e-3)
        torch.testing.assert_close(got[n], ref, rtol=2e-3, atol=max(2e-4 * scale, 1e-6), msg=n)
    for n in set(got) - set(g["grads"]):
        assert float(got[n].abs().max()) == 0.0, n


def test_backward_accumulates_for_e2e_two_passes():
    """E2E mode (train.py:397-401): both views go through `model`; gradients add up."""
    g = GOLD["e2e"]
    vq, vk = GOLD["views"]
    model = reference_encoder()
    model.load_state_dict(g["init"]["model"])
    model.train()
    # d loss / d feat from the oracle's closed form of CrossEntropy(out = fk fq^T / T, labels = arange)
    fq = g["feat_q"].clone().requires_grad_(True)
    fk = g["feat_k"].clone().requires_grad_(True)
    loss = E.nce_softmax_loss_ns(fk @ fq.t() / 0.07)
    loss.backward()
    eng = emu_engine()
    bq, bk = CpuBatch(vq), CpuBatch(vk)
    pq, bufq = eng.make_pass(model, bq, training=True, keep=g["masks"][:5].contiguous(), slot=0)
    eng.forward([pq])
    eng.backward(model, pq, bufq, fq.grad)
    pk, bufk = eng.make_pass(model, bk, training=True, keep=g["masks"][5:].contiguous(), slot=1)
    eng.forward([pk])
    eng.backward(model, pk, bufk, fk.grad, accumulate=True)
    got = _grads_after_backward(model)
    for n, ref in g["grads"].items():
        scale = max(float(ref.abs().max()), 1e-3)
        torch.testing.assert_close(got[n], ref, rtol=2e-3, atol=max(2e-4 * scale, 1e-6), msg=n)


def test_api_path_autograd_matches_golden(monkeypatch):
    """model(graph) -> contrast -> criterion -> loss.backward() exactly as train.py:389-408 spells it."""
    from gcc_amd.contrast import MemoryMoCo, NCESoftmaxLoss
    from tests.test_nce_emu import emu_nce

    g = GOLD["moco"]
    model, ema = reference_encoder(), reference_encoder()
    model.load_state_dict(g["init"]["model"])
    ema.load_state_dict(g["init"]["model_ema"])
    model._engine = ema._engine = emu_engine()
    model.train()
    _set_bn_train(ema)
    contrast = MemoryMoCo(64, None, g["K"], g["T"], use_softmax=True)
    contrast._engine = emu_nce()
    contrast.memory.copy_(g["init"]["memory"])
    masks = g["masks"].contiguous()
    monkeypatch.setattr(torch, "rand", lambda *a, **k: masks.clone())    # keep = (rand >= 0.5) reproduces the golden masks
    bq, bk = CpuBatch(GOLD["views"][0]), CpuBatch(GOLD["views"][1])
    feat_q = model(bq)
    with torch.no_grad():
        feat_k = ema(bk)
    out = contrast(feat_q, feat_k)
    loss = NCESoftmaxLoss()(out)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=1e-4, atol=1e-5)
    got = _grads_after_backward(model)
    for n, ref in g["grads"].items():
        scale = max(float(ref.abs().max()), 1e-3)
        torch.testing.assert_close(got[n], ref, rtol=2e-3, atol=max(2e-4 * scale, 1e-6), msg=n)
    # parameters the reference leaves without gradient (set2set, lin_readout) stay that way
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith(("set2set", "lin_readout")))


def test_grid_smaller_than_the_batch_walks_on():
    """gcc_gin_pass.rows_hint sizes the tile kernels' grids (round 6); a batch with more tiles than the grid has workgroups must come out
    the same: every workgroup walks on from its first tile in steps of the grid, and only the first tile uses the speculative requests
    issued with the node count.  22 graphs, 2,400 nodes = 38 tiles against the smallest grid (32 workgroups, rows_hint = 1); emulator
    tier of tests/test_encoder_gpu.py::test_tile_grid_smaller_than_the_batch_walks_on."""
    import numpy as np
    import scipy.sparse as sp

    rng = np.random.default_rng(0)
    offs, rows, cols = [0], [], []
    for _ in range(22):
        n = int(rng.integers(80, 130))
        a = np.r_[rng.integers(0, n, 3 * n), np.arange(n)]
        b = np.r_[rng.integers(0, n, 3 * n), (np.arange(n) + 1) % n]
        k = a != b
        rows += list(offs[-1] + a[k]) + list(offs[-1] + b[k])
        cols += list(offs[-1] + b[k]) + list(offs[-1] + a[k])
        offs.append(offs[-1] + n)
    N = offs[-1]
    A = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(N, N))
    A.sum_duplicates()
    A.sort_indices()
    assert N > 32 * 64 and np.diff(A.indptr).min() > 0
    view = dict(node_off=torch.tensor(offs), row_ptr=torch.from_numpy(A.indptr.astype(np.int64)),
                col_idx=torch.from_numpy(A.indices.astype(np.int64)),
                pos_undirected=torch.randn(N, 32, generator=torch.Generator().manual_seed(2)) * 0.2)
    keep = (torch.rand(5, 22, 64, generator=torch.Generator().manual_seed(3)) > 0.5).float()
    d = torch.randn(22, 64, generator=torch.Generator().manual_seed(4))
    outs = []
    for hint in (None, 1):
        torch.manual_seed(1)
        model = reference_encoder()
        model.train()
        eng = emu_engine()
        model._engine = eng
        eng.rows_hint = hint                      # None: taken from the batch (43 tiles -> 64 workgroups); 1: 32 workgroups
        p, buf = eng.make_pass(model, CpuBatch(view), training=True, keep=keep)
        eng.forward([p])
        eng.backward(model, p, buf, d)
        outs.append((buf["feat"].clone(), [q.grad.clone() for q in model.parameters() if q.grad is not None]))
    torch.testing.assert_close(outs[1][0], outs[0][0], rtol=1e-5, atol=1e-6)
    for a, b in zip(outs[1][1], outs[0][1]):      # (fp32 partial sums group differently when a workgroup walks two tiles)
        torch.testing.assert_close(a, b, rtol=0, atol=max(2e-4 * float(b.abs().max()), 1e-5))       # (a Linear bias in front of a BatchNorm: noise around 0)
