"""Parity tests proper for the GIN encoder and the MoCo/InfoNCE head on a real
MI355X: HIP kernels (through the C ABI) vs the reference-generated golden
vectors and the CPU oracle.  Tolerance: north_star asks 1e-3 rel-fp32 for
embeddings / NCE loss; the f32-MFMA path is held to 1e-4."""
import os

import pytest
import torch

from oracle import encoder as E
from tests.hipemu.emu_encoder import CpuBatch, reference_encoder

pytestmark = pytest.mark.gpu
GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_golden.pt"), weights_only=False)
TOL = dict(rtol=1e-4, atol=2e-5)


def gpu_batch(view, node_cap=None):
    b = CpuBatch(view, node_cap)
    for name in ("node_off", "edge_off", "row_ptr", "col_idx", "graph_id", "parent_nid", "pos_undirected"):
        setattr(b, name, getattr(b, name).cuda())
    return b


def _set_bn_train(model):
    model.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.train()


def _check_grads(model, ref_grads):
    got = {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None}
    assert set(ref_grads) <= set(got)
    for n, ref in ref_grads.items():
        scale = max(float(ref.abs().max()), 1e-3)
        atol = max(2e-4 * scale, 1e-6)
        if ".mlp.linears." in n and n.endswith(".bias"):
            atol = 1e-4     # a bias feeding a BatchNorm has zero gradient; both sides hold rounding noise only
        torch.testing.assert_close(got[n], ref, rtol=2e-3, atol=atol, msg=n)


def test_forward_backward_match_reference_golden():
    g = GOLD["moco"]
    model, ema = reference_encoder().cuda(), reference_encoder().cuda()
    model.load_state_dict(g["init"]["model"])
    ema.load_state_dict(g["init"]["model_ema"])
    model.train()
    _set_bn_train(ema)
    eng = model.engine()
    bq, bk = gpu_batch(GOLD["views"][0]), gpu_batch(GOLD["views"][1])
    pq, bufq = eng.make_pass(model, bq, training=True, keep=g["masks"].cuda().contiguous(), slot=0)
    pk, bufk = eng.make_pass(ema, bk, training=True, keep=None, slot=1)
    eng.forward([pq, pk])
    torch.testing.assert_close(bufq["feat"].cpu(), g["feat_q"], **TOL)
    torch.testing.assert_close(bufk["feat"].cpu(), g["feat_k"], **TOL)
    after = g["after"]["model"]
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            torch.testing.assert_close(v.cpu(), after[k], rtol=1e-4, atol=1e-5, msg=k)
    eng.backward(model, pq, bufq, g["dfeat_q"].cuda())
    _check_grads(model, g["grads"])


def test_eval_mode_matches_reference_golden():
    g = GOLD["moco"]
    model = reference_encoder().cuda()
    model.load_state_dict(g["after"]["model"])
    model.eval()
    with torch.no_grad():
        feat = model(gpu_batch(GOLD["views"][0]))
    torch.testing.assert_close(feat.cpu(), g["feat_eval"], **TOL)


def test_api_path_train_step_matches_golden(monkeypatch):
    """train.py:389-431 spelled with the drop-in modules, on the GPU."""
    from gcc_amd.contrast import MemoryMoCo, NCESoftmaxLoss

    g = GOLD["moco"]
    model, ema = reference_encoder().cuda(), reference_encoder().cuda()
    model.load_state_dict(g["init"]["model"])
    ema.load_state_dict(g["init"]["model_ema"])
    model.train()
    _set_bn_train(ema)
    contrast = MemoryMoCo(64, None, g["K"], g["T"], use_softmax=True).cuda()
    contrast.memory.copy_(g["init"]["memory"])
    masks = g["masks"].cuda().contiguous()
    monkeypatch.setattr(torch, "rand", lambda *a, **k: masks.clone())
    optimizer = torch.optim.Adam(model.parameters(), lr=0.005, betas=(0.9, 0.999), weight_decay=1e-5)
    feat_q = model(gpu_batch(GOLD["views"][0]))
    with torch.no_grad():
        feat_k = ema(gpu_batch(GOLD["views"][1]))
    out = contrast(feat_q, feat_k)
    prob = out[:, 0].mean()
    optimizer.zero_grad()
    loss = NCESoftmaxLoss()(out)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g["loss"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(prob.cpu(), g["prob"], rtol=1e-4, atol=1e-5)
    _check_grads(model, g["grads"])
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    torch.testing.assert_close(gn.cpu(), g["grad_norm"], rtol=1e-3, atol=1e-5)
    for grp in optimizer.param_groups:
        grp["lr"] = g["lr"]
    optimizer.step()
    from gcc_amd.train_step import moment_update
    moment_update(model, ema, 0.999)
    torch.testing.assert_close(contrast.memory.cpu(), g["after"]["memory"], **TOL)
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v.cpu(), g["after"]["model"][k], rtol=1e-3, atol=2e-5, msg=k)
    for k, v in ema.state_dict().items():
        torch.testing.assert_close(v.cpu(), g["after"]["model_ema"][k], rtol=1e-3, atol=2e-5, msg=k)


def test_e2e_two_passes_match_reference_golden(monkeypatch):
    from gcc_amd.contrast import NCESoftmaxLossNS, e2e_logits

    g = GOLD["e2e"]
    model = reference_encoder().cuda()
    model.load_state_dict(g["init"]["model"])
    model.train()
    masks = [g["masks"][:5].cuda().contiguous(), g["masks"][5:].cuda().contiguous()]
    monkeypatch.setattr(torch, "rand", lambda *a, **k: masks.pop(0))
    feat_q = model(gpu_batch(GOLD["views"][0]))
    feat_k = model(gpu_batch(GOLD["views"][1]))
    out = e2e_logits(feat_q, feat_k, 0.07)
    loss = NCESoftmaxLossNS()(out)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g["loss"], rtol=1e-4, atol=1e-5)
    _check_grads(model, g["grads"])


@pytest.mark.parametrize("B,K", [(5, 7), (70, 200), (256, 16384)])
def test_moco_head_vs_oracle(B, K):
    from gcc_amd.contrast import MemoryMoCo

    torch.manual_seed(B * 1000 + K)
    q = torch.nn.functional.normalize(torch.randn(B, 64), dim=1)
    k = torch.nn.functional.normalize(torch.randn(B, 64), dim=1)
    mem = E.memory_init(K, 64)
    index0 = K // 3
    ref_mem = mem.clone()
    qo = q.clone().requires_grad_(True)
    out_ref, idx_ref = E.moco_forward(ref_mem, index0, qo, k, 0.07)
    loss_ref = E.nce_softmax_loss(out_ref)
    loss_ref.backward()
    contrast = MemoryMoCo(64, None, K, 0.07, use_softmax=True).cuda()
    contrast.memory.copy_(mem)
    contrast.index = index0
    qd = q.cuda().requires_grad_(True)
    out = contrast(qd, k.cuda())
    torch.testing.assert_close(out.loss.detach().cpu(), loss_ref.detach(), rtol=1e-4, atol=1e-5)   # north_star: 1e-3
    torch.testing.assert_close(contrast.memory.cpu(), ref_mem)
    assert contrast.index == idx_ref
    torch.testing.assert_close(out.dense().cpu(), out_ref.detach(), rtol=1e-4, atol=1e-4)
    out.loss.backward()
    torch.testing.assert_close(qd.grad.cpu(), qo.grad, rtol=1e-3, atol=1e-7)


def test_long_rows_full_size_batch_vs_oracle():
    """a real sampled batch from G1-like graph (hub rows, ragged tiles): device encoder vs CPU oracle."""
    import numpy as np

    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.sampler import DeviceRWRSampler

    rp, ci = powerlaw_graph(100000, 1000000, 2)
    g = DeviceGraph(rp, ci, rw_hops=256)
    s = DeviceRWRSampler(g, 32, run_seed=5)
    q, _ = s.sample(0)
    s.check_status()
    n = q.number_of_nodes()
    torch.manual_seed(0)
    pos = torch.randn(q.parent_nid.numel(), 32, device="cuda") * 0.2
    q.pos_undirected = pos
    oracle = E.OracleGraphEncoder()
    model = reference_encoder().cuda()
    model.load_state_dict(oracle.state_dict())
    model.train()
    oracle.train()
    keep = (torch.rand(5, 32, 64) > 0.5).float()
    c = q.csr_numpy()
    ref = oracle(c["node_off"].astype(np.int64), c["row_ptr"].astype(np.int64), c["col_idx"].astype(np.int64),
                 pos[:n].cpu(), dropout_masks=keep)
    eng = model.engine()
    p, buf = eng.make_pass(model, q, training=True, keep=keep.cuda())
    eng.forward([p])
    torch.testing.assert_close(buf["feat"].cpu(), ref.detach(), rtol=1e-3, atol=1e-4)
    dfeat = torch.randn(32, 64)
    ref.backward(dfeat)
    eng.backward(model, p, buf, dfeat.cuda())
    _check_grads(model, {n: p_.grad for n, p_ in oracle.named_parameters() if p_.grad is not None})


def test_tile_grid_smaller_than_the_batch_walks_on():
    """gcc_gin_pass.rows_hint sizes the tile kernels' grids for the rows EXPECTED; a batch with more rows than that must come out the
    same (workgroups walk on from their first tile in steps of the grid).  A hint of 1 gives the smallest grid (32 workgroups): every
    workgroup walks several tiles of this ~3000-node batch, the speculative first-tile requests are used by the first one only."""
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.sampler import DeviceRWRSampler

    rp, ci = powerlaw_graph(100000, 1000000, 2)
    g = DeviceGraph(rp, ci, rw_hops=256)
    s = DeviceRWRSampler(g, 32, run_seed=6)
    q, _ = s.sample(0)
    assert q.number_of_nodes() > 32 * 64                     # more tiles than the small grid has workgroups
    torch.manual_seed(1)
    q.pos_undirected = torch.randn(q.parent_nid.numel(), 32, device="cuda") * 0.2
    keep = (torch.rand(5, 32, 64) > 0.5).float().cuda()
    dfeat = torch.randn(32, 64).cuda()
    out = []
    for hint in (None, 1, q.number_of_nodes() + 1):
        torch.manual_seed(2)
        model = reference_encoder().cuda()
        model.train()
        eng = model.engine()
        eng.rows_hint = hint
        p, buf = eng.make_pass(model, q, training=True, keep=keep)
        eng.forward([p])
        eng.backward(model, p, buf, dfeat)
        out.append((buf["feat"].clone(), [p_.grad.clone() for p_ in model.parameters() if p_.grad is not None]))
    for feat, grads in out[1:]:
        torch.testing.assert_close(feat, out[0][0], rtol=1e-5, atol=1e-6)
        for a, b in zip(grads, out[0][1]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 + 1e-5 * float(b.abs().max()))


def test_bf16_head_on_the_device_at_full_size():
    """GCC_NCE_BF16 at (B 256, K 16384): equals the oracle evaluated on bf16-rounded operands (loss, lse, dq); the distance to
    the fp32 mode is printed -- it is what rules the mode out for the 1e-3 parity bar (f32 stays the default)."""
    from gcc_amd.contrast import MemoryMoCo, NCESoftmaxLoss

    torch.manual_seed(3)
    B, K = 256, 16384
    q = torch.nn.functional.normalize(torch.randn(B, 64), dim=1)
    k = torch.nn.functional.normalize(torch.randn(B, 64), dim=1)
    mem = E.memory_init(K, 64)
    rnd = lambda t: t.to(torch.bfloat16).to(torch.float32)
    qr = rnd(q).requires_grad_(True)
    out_r, _ = E.moco_forward(rnd(mem), 0, qr, rnd(k), 0.07)
    loss_r = E.nce_softmax_loss(out_r)
    loss_r.backward()
    out_f, _ = E.moco_forward(mem.clone(), 0, q.clone(), k, 0.07)
    loss_f = E.nce_softmax_loss(out_f)
    losses = {}
    for mode in ("bf16", "f32"):
        c = MemoryMoCo(64, None, K, 0.07, use_softmax=True, nce_dtype=mode).cuda()
        c.memory.copy_(mem.cuda())
        qd = q.clone().cuda().requires_grad_(True)
        out = c(qd, k.cuda())
        loss = NCESoftmaxLoss()(out)
        loss.backward()
        losses[mode] = float(loss.detach().cpu())
        if mode == "bf16":
            torch.testing.assert_close(loss.detach().cpu(), loss_r.detach(), rtol=2e-5, atol=2e-5)
            torch.testing.assert_close(qd.grad.cpu(), qr.grad, rtol=2e-3, atol=1e-7)
        else:
            torch.testing.assert_close(loss.detach().cpu(), loss_f.detach(), rtol=1e-5, atol=1e-5)
    print("loss f32 %.6f bf16 %.6f (|diff| %.2e)" % (losses["f32"], losses["bf16"], abs(losses["f32"] - losses["bf16"])))
    assert abs(losses["f32"] - losses["bf16"]) < 2e-2
