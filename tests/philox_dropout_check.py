"""Shared body of the in-kernel Philox dropout parity test (emulator tier on CPU, device tier on the GPU).
TEST INFRASTRUCTURE ONLY."""
import torch


def philox_keep_masks(seed64, layers, B, p):
    """Host restatement of drop_mul4 (gcc_amd/csrc/encoder_common.h): one Philox4x32-10 call per 4 consecutive
    channels, counter (b * 16 + ch / 4, layer, 0xD50F, 0), key = the 64-bit seed; keep <=> (x >> 8) >= p * 2^24."""
    from oracle.sampler import py_philox4x32_10

    key = [seed64 & 0xFFFFFFFF, (seed64 >> 32) & 0xFFFFFFFF]
    keep = torch.zeros(layers, B, 64)
    thr = p * 16777216.0
    for layer in range(layers):
        for b in range(B):
            for q in range(16):
                x = py_philox4x32_10([b * 16 + q, layer, 0xD50F, 0], key)
                for u in range(4):
                    keep[layer, b, 4 * q + u] = 1.0 if float(x[u] >> 8) >= thr else 0.0
    return keep


def check_philox_dropout(make_trainer, GOLD, sync=lambda: None):
    """``make_trainer() -> (golden, MoCoTrainStep, model, ema, contrast)`` on the golden batch.  The step with
    mask_fn=None (Philox masks drawn inside the readout kernels, gin.py:202,230) must equal
    (a) the same fused step with the host-predicted masks injected -- the path the golden tests pin to the reference --
        in loss, gradient and post-step parameters: forward and backward regenerated the same mask;
    (b) oracle/encoder.py fed those masks, loss and gradients within 1e-3.
    Keep rate 0.5 within 3 sigma."""
    from gcc_amd.encoder import grad_params
    from oracle import encoder as E

    g, step, model, ema, contrast = make_trainer()
    step.mask_fn = None
    step.dropout_seed = 0x1234567
    assert model.gnn.drop.p == 0.5
    at_step = 3
    seed64 = (step.dropout_seed + at_step * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    L, B = g["masks"].shape[0], g["masks"].shape[1]
    keep = philox_keep_masks(seed64, L, B, 0.5)
    rate = float(keep.mean())
    sigma = 0.5 / (keep.numel() ** 0.5)
    assert abs(rate - 0.5) < 3 * sigma, rate
    assert (keep != g["masks"]).any()
    mem0 = contrast.memory.clone().cpu()
    out = step.step(at_step, g["lr"])
    sync()
    grad_philox = step.flat_grad.clone().cpu()
    # (a) the same step with the predicted masks injected
    g2, step2, model2, ema2, contrast2 = make_trainer()
    kd = keep.to(step2.flat_grad.device).contiguous()
    step2.mask_fn = lambda: kd
    out2 = step2.step(at_step, g["lr"])
    sync()
    torch.testing.assert_close(out["loss"].cpu(), out2["loss"].cpu(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(grad_philox, step2.flat_grad.cpu(), rtol=1e-5, atol=1e-8)
    for (k, v), (_, w) in zip(model.state_dict().items(), model2.state_dict().items()):
        torch.testing.assert_close(v.cpu(), w.cpu(), rtol=1e-5, atol=1e-7, msg=k)
    # (b) the CPU oracle with those masks
    om, oe = E.OracleGraphEncoder(), E.OracleGraphEncoder()
    om.load_state_dict(g["init"]["model"])
    oe.load_state_dict(g["init"]["model_ema"])
    om.train()
    oe.train()
    vq, vk = GOLD["views"]

    def args(v):
        return v["node_off"].long(), v["row_ptr"].long(), v["col_idx"].long(), v["pos_undirected"]

    rq = om(*args(vq), dropout_masks=keep)
    with torch.no_grad():
        rk = oe(*args(vk))
    rout, _ = E.moco_forward(mem0, 0, rq, rk, g["T"])
    rloss = E.nce_softmax_loss(rout)
    rloss.backward()
    torch.testing.assert_close(out["loss"].reshape(()).cpu(), rloss.detach(), rtol=1e-3, atol=1e-5)
    full = {id(p): n for n, p in model.named_parameters()}
    ref = dict(om.named_parameters())
    # gcc_adam_step clips the flat gradient in place (clip_grad_norm_ does the same to .grad, train.py:409)
    gn = float(torch.as_tensor(out["grad_norm"]).reshape(()).cpu())
    gn_ref = float(torch.sqrt(sum((p.grad ** 2).sum() for p in om.parameters() if p.grad is not None)))
    assert abs(gn - gn_ref) <= 1e-3 * gn_ref, (gn, gn_ref)
    coef = min(1.0, step.clip_norm / (gn_ref + 1e-6))
    off, seen = 0, 0
    for _, _, p in grad_params(model):
        got = grad_philox[off:off + p.numel()].view_as(p)
        off += p.numel()
        gref = ref[full[id(p)]].grad * coef
        scale = float(gref.abs().max())                 # the golden gradient tests' tolerance (tests/test_encoder_emu.py)
        torch.testing.assert_close(got, gref, rtol=2e-3, atol=max(2e-4 * scale, 1e-6), msg=full[id(p)])
        seen += 1
    assert seen > 20
