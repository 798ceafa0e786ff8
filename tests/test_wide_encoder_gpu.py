"""Device tier of tests/test_wide_encoder_emu.py: --hidden-size above 64 (train.py:93) on a device-sampled batch with the
device positional embedding -- GraphEncoder forward (training mode) + backward and the MoCo head at width 128 / 256 against
oracle/encoder.py built at the same width; and train.py --hidden-size 128 end to end (API path: any-width kernels, torch
Adam), checkpoint with the reference's shapes, generate.py on it."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden,B,hops", [(128, 32, 64), (256, 64, 256)])
def test_wide_encoder_and_head_on_the_device_vs_oracle(hidden, B, hops, monkeypatch):
    from gcc_amd.contrast import MemoryMoCo, NCESoftmaxLoss
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from oracle import encoder as E
    from tests.headline_step_check import view_arrays
    from tests.test_wide_encoder_emu import wide_encoder

    rp, ci = powerlaw_graph(100_000, 1_000_000, 3)
    graph = DeviceGraph(rp, ci, rw_hops=hops, device="cuda:0")
    K = 1024
    torch.manual_seed(hidden)
    model, ema = wide_encoder(hidden, hidden).cuda(), wide_encoder(hidden, hidden).cuda()
    ema.load_state_dict(model.state_dict())
    om, oe = E.OracleGraphEncoder(node_hidden_dim=hidden, output_dim=hidden), E.OracleGraphEncoder(node_hidden_dim=hidden, output_dim=hidden)
    om.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    oe.load_state_dict(om.state_dict())
    contrast = MemoryMoCo(hidden, None, K, 0.07, use_softmax=True).cuda()
    mem = contrast.memory.detach().cpu().clone()
    mem0 = mem.clone()
    smp = DeviceRWRSampler(graph, B, run_seed=4)
    pe = DevicePosEmb(B, smp.node_cap, 32, device="cuda:0", seed=4)
    q, k = smp.sample(0)
    pe(q)
    pe(k)
    model.train()
    om.train()
    ema.eval()
    oe.eval()
    for m in list(ema.modules()) + list(oe.modules()):             # train.py:357-365
        if isinstance(m, torch.nn.BatchNorm1d):
            m.train()
    keep = (torch.rand(5, B, hidden) >= 0.5).float()
    monkeypatch.setattr(torch, "rand", lambda *a, **kw: keep.clone().to(kw.get("device", "cpu")))
    feat_q = model(q)
    with torch.no_grad():
        feat_k = ema(k)
    out = contrast(feat_q, feat_k)
    loss = NCESoftmaxLoss()(out)
    loss.backward()
    torch.cuda.synchronize()
    (aq, pos_q), (ak, pos_k) = view_arrays(q), view_arrays(k)
    rq = om(*aq, pos_q, dropout_masks=keep)
    with torch.no_grad():
        rk = oe(*ak, pos_k)
    rout, _ = E.moco_forward(mem, 0, rq, rk, 0.07)
    rloss = E.nce_softmax_loss(rout)
    rloss.backward()
    torch.testing.assert_close(feat_q.detach().cpu(), rq.detach(), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(feat_k.cpu(), rk, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(loss.detach().cpu(), rloss.detach(), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(contrast.memory.cpu(), mem, rtol=0, atol=1e-6)
    # gradients: the bar proper is the oracle run in float64 (exact arithmetic for this purpose) at north_star's 1e-3 of the tensor's
    # largest entry; the fp32 oracle's own sums over thousands of nodes are ~1e-3 away from it themselves
    o64 = E.OracleGraphEncoder(node_hidden_dim=hidden, output_dim=hidden).double()
    o64.load_state_dict({k: (v.cpu().double() if v.dtype.is_floating_point else v.cpu()) for k, v in ema.state_dict().items()})   # (= the initial weights)
    o64.train()
    f64 = o64(*aq, pos_q.double(), dropout_masks=keep.double())
    out64, _ = E.moco_forward(mem0.double(), 0, f64, rk.double(), 0.07)
    E.nce_softmax_loss(out64).backward()
    ref64, ref32 = dict(o64.named_parameters()), dict(om.named_parameters())
    worst, worst32, table = 0.0, 0.0, []
    for name, p in model.named_parameters():
        if ref64[name].grad is None:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name
            continue
        g64 = ref64[name].grad.float()
        scale = max(float(g64.abs().max()), 1e-3)
        # the bar: north_star's 1e-3 of the tensor's largest entry, against the float64 run, no allowance for fp32 (round 5 accepted three
        # times torch's own fp32 error; with the weight gradients' slabs added up in fp64 the device sits where torch's fp32 pass sits:
        # 5e-4 .. 7e-4 at both widths, the two within 1e-6 of EACH OTHER -- profiles/r6_wide_gradient_errors.txt)
        err32 = float((ref32[name].grad - g64).abs().max())
        torch.testing.assert_close(p.grad.cpu(), g64, rtol=0, atol=1e-3 * scale, msg=lambda m, name=name: f"{name} vs float64 oracle: {m}")
        worst = max(worst, float((p.grad.cpu() - g64).abs().max()) / scale)
        worst32 = max(worst32, float((ref32[name].grad - g64).abs().max()) / scale)
        table.append((float((p.grad.cpu() - g64).abs().max()) / scale, err32 / scale, name))
    for dev_err, o32, name in sorted(table, reverse=True)[:8]:
        print(f"   {name:60s} device {dev_err:.2e}   fp32 oracle {o32:.2e}   (of the tensor's largest entry, vs float64)")
    print(f"hidden {hidden}: {int(aq[0][-1])} nodes, loss {float(loss):.5f} (oracle {float(rloss):.5f}), worst gradient error / scale vs float64: "
          f"device {worst:.2e}, fp32 oracle {worst32:.2e}")


def test_train_py_with_hidden_size_128_then_generate(tmp_path):
    import io
    from contextlib import redirect_stdout

    import generate
    import train
    from tests.test_train_main_gpu import _corpus

    corpus, gs = _corpus(tmp_path)
    argv = ["--exp", "wide", "--model-path", str(tmp_path / "s"), "--tb-path", str(tmp_path / "t"), "--gpu", "0", "--moco", "--nce-k", "256",
            "--hidden-size", "128", "--batch-size", "32", "--num-workers", "2", "--num-copies", "1", "--num-samples", "256",
            "--rw-hops", "64", "--dgl-file", corpus, "--epochs", "2", "--print-freq", "4", "--tb-freq", "1000"]
    args = train.parse_option(argv)
    args.gpu = args.gpu[0]
    buf = io.StringIO()
    with redirect_stdout(buf):
        loss = train.main(args)
    vals = [float(l.split("loss ")[1].split(" ")[0]) for l in buf.getvalue().splitlines() if l.startswith("Train:")]
    assert len(vals) == 8 and all(np.isfinite(v) and 0.0 < v < 7.0 for v in vals), vals      # 2 epochs x 16 steps / 4
    assert np.isfinite(loss) and "_hid_128_" in os.path.basename(args.model_folder)
    ckpt = torch.load(os.path.join(args.model_folder, "current.pth"), map_location="cpu", weights_only=False)
    assert ckpt["contrast"]["memory"].shape == (256, 128)
    assert ckpt["model"]["gnn.ginlayers.1.apply_func.mlp.linears.0.weight"].shape == (128, 128)
    assert ckpt["model"]["gnn.linears_prediction.0.weight"].shape == (128, 49)
    rp, ci = gs[2]
    npz = tmp_path / "g.npz"
    np.savez(npz, row_ptr=rp, col_idx=ci)
    a = types.SimpleNamespace(load_path=os.path.join(args.model_folder, "current.pth"), dataset="toy", gpu=0, edgelist=None,
                              nodelabel=None, graph_npz=str(npz), graphs_npz=None, tudataset=None, edge_multiplicity=2, batch_size=64)
    generate.main(a)
    emb = np.load(os.path.join(args.model_folder, "toy.npy"))
    assert emb.shape == (len(rp) - 1, 128) and np.isfinite(emb).all()
