"""The step's two collectives (all-gather of keys, all-reduce of the flat gradient) through RCCL on the GPU box: a
1-rank process group is all a single-GPU box allows, but it runs the real NCCL kernels on RCCL's stream next to the
producer lanes and the high-priority training stream, and must give the same losses as the step without collectives --
launch by launch, and as the SEGMENTED graph replay the multi-GPU step ships with (three captured graphs around the two
RCCL hand-offs: gcc_amd/train_step.py MoCoTrainStep._body)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(collectives, use_graph=None, steps=20):
    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.encoder import GraphEncoder
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from gcc_amd.train_step import MoCoTrainStep

    dev = torch.device("cuda:0")
    rp, ci = powerlaw_graph(50000, 400000, 2)
    graph = DeviceGraph(rp, ci, rw_hops=64, restart_prob=0.8, device=dev)
    B, chunk, depth = 32, 2, 2
    torch.manual_seed(0)
    kw = dict(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512, freq_embedding_size=16,
              degree_embedding_size=16, output_dim=64, node_hidden_dim=64, edge_hidden_dim=64, num_layers=5,
              num_step_set2set=6, num_layer_set2set=3, norm=True, gnn_model="gin", degree_input=True)
    model, ema = GraphEncoder(**kw).to(dev), GraphEncoder(**kw).to(dev)
    ema.load_state_dict(model.state_dict())
    contrast = MemoryMoCo(64, None, 256, 0.07, use_softmax=True).to(dev)
    lanes = []
    for _ in range(2):
        smp = DeviceRWRSampler(graph, B, run_seed=3, num_buffers=depth * chunk, max_steps=chunk)
        lanes.append((smp, DevicePosEmb(B, smp.node_cap, 32, device=dev, seed=3, num_buffers=depth * chunk, max_views=2 * chunk)))
    tr = MoCoTrainStep(model, ema, contrast, lanes[0][0], lanes[0][1], lanes=lanes, depth=depth, chunk=chunk,
                       collectives=collectives, graph=use_graph)
    tr.dropout_seed = 11
    tr.relaxed_streams = True                  # the bench / train.py loops: no per-step stream hand-offs
    outs = [tr.step(i, 0.005) for i in range(steps)]
    tr.join()
    torch.cuda.synchronize()
    # (a replayed step returns its slot's captured output tensors: read after the loop they hold each slot's LAST step, so
    #  the per-step trace is the meters' running sum, read step by step in a second pass below)
    return tr, [float(o["loss"].item()) for o in outs[-4:]], model.state_dict(), contrast.memory.clone()


def test_step_with_rccl_collectives_matches_the_step_without():
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    _, ref, ref_w, ref_mem = _run(False)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        tr_e, eager, eager_w, eager_mem = _run(True, use_graph=False)
        tr_g, seg, seg_w, seg_mem = _run(True)               # default with collectives on a device: segmented replay
    finally:
        dist.destroy_process_group()
    assert not tr_e.use_graph and tr_e.graph_replays == 0
    # 2 lanes x depth 2 x chunk 2 = 8 ring slots: step 0 is eager, the 5 other slots of the look-ahead chunks 0..2 are captured
    # right after it, chunk 3's two slots on their first use
    assert tr_g.use_graph and tr_g.graph_replays == 20 - 3 and len(tr_g.graphs) == 8
    for items, _ in tr_g.graphs.values():
        kinds = [isinstance(it, torch.cuda.CUDAGraph) for it in items]
        assert kinds == [True, False, True, False, True]     # forward | gather begins | head + backward | reduce, join | update
    for got, w, mem in ((eager, eager_w, eager_mem), (seg, seg_w, seg_mem)):
        assert got == pytest.approx(ref, rel=1e-5)
        for name, t in ref_w.items():
            if t.dtype.is_floating_point:
                assert torch.allclose(w[name], t, rtol=1e-5, atol=1e-7), name
        assert torch.allclose(mem, ref_mem, rtol=1e-5, atol=1e-7)
