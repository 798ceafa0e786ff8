"""Device tier of tests/test_hidden_size_emu.py: --hidden-size 32 (the GraphEncoder signature's own default,
graph_encoder.py:44-63) and 48 through the fused MoCo step on a device-sampled batch, against oracle/encoder.py built at the
same width, with graph replay on (the trainer's default with prefetch)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden", [32, 48])
def test_fused_moco_step_narrow_model_on_the_device(hidden):
    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from gcc_amd.train_step import MoCoTrainStep
    from oracle import encoder as E
    from tests.headline_step_check import check_moco_step
    from tests.test_hidden_size_emu import narrow_encoder

    rp, ci = powerlaw_graph(100_000, 1_000_000, 2)
    graph = DeviceGraph(rp, ci, rw_hops=128, device="cuda:0")
    B, K = 64, 1024
    torch.manual_seed(hidden)
    model, ema = narrow_encoder(hidden, hidden).cuda(), narrow_encoder(hidden, hidden).cuda()
    ema.load_state_dict(model.state_dict())
    contrast = MemoryMoCo(hidden, None, K, 0.07, use_softmax=True).cuda()
    smp = DeviceRWRSampler(graph, B, run_seed=1, num_buffers=4, max_steps=2)
    pe = DevicePosEmb(B, smp.node_cap, 32, device="cuda:0", seed=1, num_buffers=4, max_views=4)
    tr = MoCoTrainStep(model, ema, contrast, smp, pe, depth=2, chunk=2, prefetch=True)
    assert tr.use_graph
    ref_shapes = {k: tuple(v.shape) for k, v in E.OracleGraphEncoder(node_hidden_dim=hidden, output_dim=hidden).state_dict().items()}
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == ref_shapes
    tr.dropout_seed = 3
    for i in range(6):                                   # graph capture + replays with in-kernel dropout first
        tr.step(i, 0.005)
    assert tr.graph_replays >= 2
    masks = (torch.rand(5, B, 64) >= 0.5).float().cuda().contiguous()
    rep = check_moco_step(tr, model, ema, contrast, 0.004, masks, sync=torch.cuda.synchronize, step_id=6)
    rep.pop("_graphs")
    assert tr.check_status(strict_posemb=True) == 0
    assert tuple(contrast.memory.shape) == (K, hidden) and rep["loss_rel_err"] < 1e-3
    print(f"hidden {hidden} fused step vs oracle:", rep)


def test_api_path_narrow_model_on_the_device():
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from oracle import encoder as E
    from tests.headline_step_check import view_arrays
    from tests.test_hidden_size_emu import narrow_encoder

    rp, ci = powerlaw_graph(50_000, 500_000, 3)
    graph = DeviceGraph(rp, ci, rw_hops=64, device="cuda:0")
    B, hidden, out = 32, 32, 16
    torch.manual_seed(0)
    model = narrow_encoder(hidden, out).cuda()
    oracle = E.OracleGraphEncoder(node_hidden_dim=hidden, output_dim=out)
    oracle.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    smp = DeviceRWRSampler(graph, B, run_seed=2)
    pe = DevicePosEmb(B, smp.node_cap, 32, device="cuda:0", seed=2)
    q, _ = smp.sample(0)
    pe(q)
    model.eval()
    oracle.eval()
    with torch.no_grad():
        feat = model(q)
    args, pos = view_arrays(q)
    with torch.no_grad():
        ref = oracle(*args, pos)
    assert tuple(feat.shape) == (B, out)
    torch.testing.assert_close(feat.cpu(), ref, rtol=1e-3, atol=1e-4)
