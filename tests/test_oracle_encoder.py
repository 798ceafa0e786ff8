"""Pins oracle/encoder.py against tests/golden/encoder_golden.pt, which was
produced by executing the reference's own gin.py / graph_encoder.py /
memory_moco.py / criterions.py (make_encoder_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import encoder as E

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_golden.pt"), weights_only=False)
TOL = dict(rtol=1e-5, atol=1e-6)


def _set_bn_train(model):
    model.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.train()


def test_state_dict_keys_and_shapes_match_reference():
    enc = E.OracleGraphEncoder()
    got = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert got == GOLD["state_dict_shapes"]
    assert [n for n, _ in enc.named_parameters()] == GOLD["param_names"]
    assert sum(p.numel() for p in enc.parameters()) == 190544


def test_moco_step_matches_reference():
    g = GOLD["moco"]
    vq, vk = GOLD["views"]
    model, ema = E.OracleGraphEncoder(), E.OracleGraphEncoder()
    model.load_state_dict(g["init"]["model"])
    ema.load_state_dict(g["init"]["model_ema"])
    model.train()
    _set_bn_train(ema)
    feat_q, all_q = model(vq["node_off"], vq["row_ptr"], vq["col_idx"], vq["pos_undirected"],
                          dropout_masks=g["masks"], return_all_outputs=True)
    with torch.no_grad():
        feat_k = ema(vk["node_off"], vk["row_ptr"], vk["col_idx"], vk["pos_undirected"])
    torch.testing.assert_close(feat_q, g["feat_q"], **TOL)
    torch.testing.assert_close(feat_k, g["feat_k"], **TOL)
    for a, b in zip(all_q, g["all_outputs_q"]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-4)
    memory = g["init"]["memory"].clone()
    out, index = E.moco_forward(memory, 0, feat_q, feat_k, g["T"])
    torch.testing.assert_close(out, g["out"], rtol=1e-5, atol=1e-5)
    loss = E.nce_softmax_loss(out)
    torch.testing.assert_close(loss, g["loss"], **TOL)
    torch.testing.assert_close(out[:, 0].mean(), g["prob"], **TOL)
    opt = torch.optim.Adam(model.parameters(), lr=0.005, betas=(0.9, 0.999), weight_decay=1e-5)
    opt.zero_grad()
    feat_q.retain_grad()
    loss.backward()
    torch.testing.assert_close(feat_q.grad, g["dfeat_q"], rtol=1e-5, atol=1e-7)
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(g["grads"])
    for n in grads:
        torch.testing.assert_close(grads[n], g["grads"][n], rtol=1e-4, atol=1e-5, msg=n)
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    torch.testing.assert_close(torch.as_tensor(gn), g["grad_norm"], rtol=1e-5, atol=1e-6)
    for grp in opt.param_groups:
        grp["lr"] = g["lr"]
    opt.step()
    E.moment_update(model, ema, 0.999)
    torch.testing.assert_close(memory, g["after"]["memory"], **TOL)
    assert index == g["after"]["index"]
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v, g["after"]["model"][k], rtol=1e-5, atol=1e-6, msg=k)
    for k, v in ema.state_dict().items():
        torch.testing.assert_close(v, g["after"]["model_ema"][k], rtol=1e-5, atol=1e-6, msg=k)
    model.eval()
    with torch.no_grad():
        fe = model(vq["node_off"], vq["row_ptr"], vq["col_idx"], vq["pos_undirected"])
    torch.testing.assert_close(fe, g["feat_eval"], **TOL)


def test_e2e_step_matches_reference():
    g = GOLD["e2e"]
    vq, vk = GOLD["views"]
    model = E.OracleGraphEncoder()
    model.load_state_dict(g["init"]["model"])
    model.train()
    fq = model(vq["node_off"], vq["row_ptr"], vq["col_idx"], vq["pos_undirected"], dropout_masks=g["masks"][:5])
    fk = model(vk["node_off"], vk["row_ptr"], vk["col_idx"], vk["pos_undirected"], dropout_masks=g["masks"][5:])
    torch.testing.assert_close(fq, g["feat_q"], **TOL)
    torch.testing.assert_close(fk, g["feat_k"], **TOL)
    out = fk @ fq.t() / 0.07
    loss = E.nce_softmax_loss_ns(out)
    torch.testing.assert_close(loss, g["loss"], **TOL)
    loss.backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            torch.testing.assert_close(p.grad, g["grads"][n], rtol=1e-4, atol=1e-5, msg=n)


def test_warmup_linear_matches_reference_points():
    # gcc/utils/misc.py:5-10 evaluated by the reference inside make_encoder_golden.py
    assert E.warmup_linear(3 / 7500.0, 0.1) * 0.005 == pytest.approx(GOLD["moco"]["lr"])
    assert E.warmup_linear(0.1, 0.1) == pytest.approx(1.0)
    assert E.warmup_linear(1.0, 0.1) == 0
