"""--hidden-size 128 / 256 against vectors produced by EXECUTING the reference (tests/golden/make_encoder_wide_golden.py:
the reference's GraphEncoder + MemoryMoCo + NCESoftmaxLoss + clip + Adam + moment_update at those widths, and the same modules in
float64 for the exact gradients).  One MoCo step through the any-width API path (GraphEncoder.forward -> csrc/ginx.hip,
MemoryMoCo -> gcc_ncex_forward, torch.optim.Adam as train.py's wide path uses it); shared by the emulator tier
(tests/test_wide_golden_emu.py) and the device tier (tests/test_wide_golden_gpu.py).  The bar is north_star's: 1e-3 of the
tensor's largest entry for gradients (against the float64 run), 1e-3 relative for embeddings / logits / loss."""
import os

import torch

from tests.golden import wide_init

GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "encoder_wide_golden.pt")
_gold = None


def gold():
    global _gold
    if _gold is None:
        _gold = torch.load(GOLD_PATH, weights_only=False)
    return _gold


def moment_update(model, model_ema, m):          # train.py:169-172
    for p1, p2 in zip(model.parameters(), model_ema.parameters()):
        p2.data.mul_(m).add_(p1.detach().data, alpha=1 - m)


def run_moco_step(hidden, device, to_batch, monkeypatch, gin_engine=None, nce_engine=None):
    """Returns the worst gradient error relative to the tensor's scale (vs the reference's float64 run)."""
    from gcc_amd.contrast import MemoryMoCo, NCESoftmaxLoss
    from tests.test_wide_encoder_emu import wide_encoder

    G = gold()
    c = G["cases"][hidden]
    model, ema = wide_init.fill_(wide_encoder(hidden, hidden), 0).to(device), wide_init.fill_(wide_encoder(hidden, hidden), 1).to(device)
    contrast = MemoryMoCo(hidden, None, c["K"], c["T"], use_softmax=True)
    with torch.no_grad():
        contrast.memory.copy_(wide_init.tensor_for("contrast.memory", contrast.memory) * c["memory0_scale"])
    contrast = contrast.to(device)
    if gin_engine is not None:
        model._wide_engine, ema._wide_engine = gin_engine(), gin_engine()
    if nce_engine is not None:
        contrast._engine = nce_engine()
    q, k = to_batch(G["views"][0]), to_batch(G["views"][1])
    model.train()                                                        # train.py:357-365
    ema.eval()
    for m in ema.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.train()
    masks = c["masks"]
    monkeypatch.setattr(torch, "rand", lambda *a, **kw: masks.clone().to(kw.get("device", "cpu")))      # the API path draws its dropout masks here
    optimizer = torch.optim.Adam(model.parameters(), lr=0.005, betas=(0.9, 0.999), weight_decay=1e-5)
    feat_q, pooled = model(q, return_all_outputs=True)                   # train.py:389
    with torch.no_grad():
        feat_k = ema(k)                                                  # train.py:390-391
    out = contrast(feat_q, feat_k)                                       # train.py:393
    loss = NCESoftmaxLoss()(out)                                         # train.py:407
    optimizer.zero_grad()
    feat_q.retain_grad()
    loss.backward()
    cpu = lambda t: t.detach().cpu()
    torch.testing.assert_close(cpu(feat_q), c["feat_q"], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(cpu(feat_k), c["feat_k"], rtol=1e-3, atol=2e-5)
    for a, b in zip(pooled, c["all_outputs_q"]):
        torch.testing.assert_close(cpu(a), b, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(cpu(out.dense()), c["out"], rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(cpu(loss), c["loss"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(cpu(out.prob), c["prob"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(cpu(feat_q.grad), c["dfeat_q"], rtol=1e-3, atol=1e-3 * float(c["dfeat_q"].abs().max()))
    worst = 0.0
    named = dict(model.named_parameters())
    for name, g64 in c["grads64"].items():
        scale = max(float(g64.abs().max()), 1e-3)
        got = cpu(named[name].grad)
        err = float((got - g64).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 1e-3, f"hidden {hidden}: d {name} is {err:.2e} of the tensor's largest entry away from the reference's float64 run"
        if name in c["grads"]:                   # and as close to the reference's own fp32 run as that run is to float64, plus the bar
            err32 = float((got - c["grads"][name]).abs().max()) / scale
            assert err32 <= 1e-3 + c["ref_fp32_vs_f64"], f"hidden {hidden}: d {name} vs the reference's fp32 run: {err32:.2e}"
    for name, p in named.items():                # parameters without a gradient in the reference have none here (set2set, lin_readout)
        if name not in c["grads64"]:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)         # train.py:409
    torch.testing.assert_close(cpu(gn).float(), c["grad_norm"].float(), rtol=1e-3, atol=0)
    for g in optimizer.param_groups:
        g["lr"] = c["lr"]
    optimizer.step()
    moment_update(model, ema, 0.999)                                     # train.py:430-431
    for sd_name, mod in (("model", model), ("model_ema", ema)):
        sd = mod.state_dict()
        for key, ref in c["after"][sd_name].items():
            if ref.dtype.is_floating_point:
                # post-Adam: the first step moves every weight by ~lr * sign(g): compare the MOVE, not only the value
                torch.testing.assert_close(cpu(sd[key]), ref, rtol=1e-4, atol=2e-6, msg=lambda m, key=key, n=sd_name: f"{n}.{key}: {m}")
            else:
                assert int(sd[key]) == int(ref), (sd_name, key)
    torch.testing.assert_close(cpu(contrast.memory), c["after"]["memory"], rtol=1e-3, atol=2e-5)
    assert contrast.index == c["after"]["index"]
    return worst
