"""--hidden-size below 64 (train.py:93; GraphEncoder(node_hidden_dim, output_dim), graph_encoder.py:44-63; the
constructor's own default is 32): the model runs on the 64-channel kernels as the prefix of zero-padded blocks
(gcc_amd/encoder.py: GraphEncoder.ensure_padded).  API path and fused steps against oracle/encoder.py built with the same
widths; state_dict() keeps the reference's shapes.  Emulator tier; the device tier is tests/test_hidden_size_gpu.py."""
import numpy as np
import pytest
import torch

from gcc_amd.contrast import MemoryMoCo, NCESoftmaxLoss
from gcc_amd.encoder import GraphEncoder
from gcc_amd.train_step import E2ETrainStep, MoCoTrainStep
from oracle import encoder as E
from tests.headline_step_check import check_e2e_step, check_moco_step
from tests.hipemu.emu_encoder import emu_engine
from tests.test_headline_step_emu import B, OracleSampler
from tests.test_nce_emu import emu_nce


def narrow_encoder(hidden, out):
    return GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                        freq_embedding_size=16, degree_embedding_size=16, output_dim=out, node_hidden_dim=hidden,
                        edge_hidden_dim=hidden, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                        gnn_model="gin", degree_input=True)


@pytest.mark.parametrize("hidden,out", [(32, 32), (30, 20), (64, 16)])
def test_api_path_forward_backward_vs_oracle(hidden, out, monkeypatch):
    torch.manual_seed(hidden * 100 + out)
    model = narrow_encoder(hidden, out)
    oracle = E.OracleGraphEncoder(node_hidden_dim=hidden, output_dim=out)
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v.shape) for k, v in oracle.state_dict().items()}
    oracle.load_state_dict(model.state_dict())
    model._engine = emu_engine()
    model.train()
    oracle.train()
    q, _ = OracleSampler().views
    keep = (torch.rand(5, B, 64) >= 0.5).float()
    monkeypatch.setattr(torch, "rand", lambda *a, **k: keep.clone())       # the API path draws its dropout masks here
    feat, pooled = model(q, return_all_outputs=True)
    assert tuple(feat.shape) == (B, out) and all(tuple(t.shape) == (B, hidden) for t in pooled)
    n = q.n
    args = (q.node_off.long(), q.row_ptr[: n + 1].long(), q.col_idx.long(), q.pos_undirected[:n])
    ref, ref_pooled = oracle(*args, dropout_masks=keep[:, :, :out], return_all_outputs=True)
    torch.testing.assert_close(feat, ref, rtol=1e-4, atol=2e-5)
    for a, b in zip(pooled, ref_pooled):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    d = torch.randn(B, out)
    feat.backward(d)
    ref.backward(d)
    refg = dict(oracle.named_parameters())
    for name, p in model.named_parameters():
        if refg[name].grad is None:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name
            continue
        assert p.grad.shape == p.shape
        scale = max(float(refg[name].grad.abs().max()), 1e-3)
        torch.testing.assert_close(p.grad, refg[name].grad, rtol=2e-3, atol=max(2e-4 * scale, 1e-4 if name.endswith("bias") else 5e-6), msg=lambda m, name=name: f"{name}: {m}")
    # the state survives a save / load round trip with the reference's shapes, and eval mode agrees too
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    m2 = narrow_encoder(hidden, out)
    m2.load_state_dict(sd)
    m2._engine = emu_engine()
    m2.eval()
    oracle.eval()
    with torch.no_grad():
        torch.testing.assert_close(m2(q), oracle(*args), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("hidden", [32, 48])
def test_fused_moco_step_with_a_narrow_model_vs_oracle(hidden):
    torch.manual_seed(hidden)
    model, ema = narrow_encoder(hidden, hidden), narrow_encoder(hidden, hidden)
    ema.load_state_dict(model.state_dict())
    model._engine = ema._engine = emu_engine()
    contrast = MemoryMoCo(hidden, None, 96, 0.07, use_softmax=True)
    assert tuple(contrast.memory.shape) == (96, hidden)
    contrast._engine = emu_nce()
    tr = MoCoTrainStep(model, ema, contrast, OracleSampler(), posemb=lambda gr: gr, prefetch=False)
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in E.OracleGraphEncoder(node_hidden_dim=hidden, output_dim=hidden).state_dict().items()}
    tr.dropout_seed = 3
    tr.step(0, 0.005)                                        # one ordinary step first (in-kernel dropout)
    masks = (torch.rand(5, B, 64) >= 0.5).float().contiguous()
    rep = check_moco_step(tr, model, ema, contrast, 0.004, masks, step_id=1)
    assert rep["loss_rel_err"] < 1e-3 and tuple(contrast.memory.shape) == (96, hidden)


def test_fused_e2e_step_with_a_narrow_model_vs_oracle():
    torch.manual_seed(7)
    model = narrow_encoder(32, 32)
    model._engine = emu_engine()
    tr = E2ETrainStep(model, OracleSampler(), posemb=lambda gr: gr, prefetch=False, engine=emu_nce())
    mq = (torch.rand(5, B, 64) >= 0.5).float().contiguous()
    mk = (torch.rand(5, B, 64) >= 0.5).float().contiguous()
    check_e2e_step(tr, model, 0.005, mq, mk)


def test_api_path_moco_head_with_narrow_features():
    """MemoryMoCo(inputSize=32): logits, loss, dq and the enqueue against the oracle; the queue keeps its [K, 32] shape."""
    torch.manual_seed(5)
    Bq, K, d = 10, 40, 32
    q = torch.nn.functional.normalize(torch.randn(Bq, d), dim=1).requires_grad_(True)
    k = torch.nn.functional.normalize(torch.randn(Bq, d), dim=1)
    contrast = MemoryMoCo(d, None, K, 0.07, use_softmax=True)
    contrast._engine = emu_nce()
    mem0 = contrast.memory.clone()
    qo = q.detach().clone().requires_grad_(True)
    ref_mem = mem0.clone()
    out_ref, idx = E.moco_forward(ref_mem, 0, qo, k, 0.07)
    E.nce_softmax_loss(out_ref).backward()
    out = contrast(q, k)
    loss = NCESoftmaxLoss()(out)
    loss.backward()
    torch.testing.assert_close(out.dense(), out_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(q.grad, qo.grad, rtol=1e-3, atol=1e-7)
    torch.testing.assert_close(contrast.memory, ref_mem)
    assert tuple(contrast.state_dict()["memory"].shape) == (K, d) and contrast.index == idx
