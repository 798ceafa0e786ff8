"""The BASELINE config pinned to the oracle ON THE DEVICE: one fused MoCo step (`MoCoTrainStep`, the code bench.py
times) at configs[1] -- G1 (1M nodes / 10M edges), bsz 256, rw_hops 256, restart 0.8, K 16384 -- sampled by the device
sampler, positional embedding by the device eigensolvers, against (i) oracle/sampler_oracle.c bit for bit (node ids,
batched CSR of both views) and (ii) oracle/encoder.py fed the SAME CSR + positional embedding + dropout masks: feat_q,
feat_k, loss, prob, grad-norm, every gradient, post-Adam weights, EMA weights, BatchNorm running statistics and queue
rows at north_star's 1e-3.  Same for the fused E2E step (configs[0] shape at bsz 256, train.py:396-417).

This is the 768-workgroup tile walk, the 32-replica fp64 statistics under contention, gin_wgrad's slabs at N ~ 25 k and
the head at K 16384 against the CPU restatement, not GPU against GPU.  Reference: train.py:378-434."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, K, HOPS, RESTART, RUN_SEED = 256, 16384, 256, 0.8, 0
_G1 = {}


def _g1():
    if not _G1:
        from gcc_amd.graph import DeviceGraph
        from gcc_amd.graphgen import powerlaw_graph

        rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
        _G1.update(rp=rp, ci=ci, graph=DeviceGraph(rp, ci, rw_hops=HOPS, restart_prob=RESTART, device="cuda:0", validate=False, trusted=True))
    return _G1


def _lane(graph):
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler

    smp = DeviceRWRSampler(graph, B, run_seed=RUN_SEED, num_buffers=2)
    return smp, DevicePosEmb(B, smp.node_cap, 32, device="cuda:0", seed=RUN_SEED, num_buffers=2, max_views=2)


def _encoder():
    from tests.hipemu.emu_encoder import reference_encoder

    return reference_encoder().cuda()


def _sampler_bit_exact(g1, out, first_id):
    """both views of the step's batch vs the C oracle (bit-exact subgraph node-id sets / batched CSR, north_star)."""
    from oracle import sampler as O

    rp, ci = g1["rp"], g1["ci"]
    c = O.COracle()
    seeds = c.draw_seeds(O.seed_cdf(rp), RUN_SEED, first_id, B)
    L = O.max_nodes_table(int(np.diff(rp).max()), HOPS, RESTART)[np.diff(rp)[seeds]]
    for view, gb in enumerate((out["graph_q"], out["graph_k"])):
        ref = c.sample_batch(rp, ci, seeds, L, view, RUN_SEED, first_id, O.restart_threshold(RESTART), threads=c.max_threads())
        got = gb.csr_numpy()
        for key in ("node_off", "parent_nid", "row_ptr", "col_idx"):
            assert np.array_equal(got[key][: len(ref[key])], ref[key]), (view, key)


def test_moco_step_at_the_baseline_config_vs_oracle():
    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.train_step import MoCoTrainStep
    from tests.headline_step_check import check_moco_step

    g1 = _g1()
    torch.manual_seed(0)
    model, ema = _encoder(), _encoder()
    ema.load_state_dict(model.state_dict())
    contrast = MemoryMoCo(64, None, K, 0.07, use_softmax=True).cuda()
    smp, pe = _lane(g1["graph"])
    tr = MoCoTrainStep(model, ema, contrast, smp, pe, prefetch=False)
    masks = (torch.rand(5, B, 64) >= 0.5).float().cuda().contiguous()
    rep = check_moco_step(tr, model, ema, contrast, 0.005, masks, sync=torch.cuda.synchronize)
    assert tr.check_status(strict_posemb=True) == 0
    assert rep["nodes_q"] > 10 * B and rep["nodes_k"] > 10 * B, rep          # a real C2 batch (~ 25 k nodes per view)
    q, k = rep.pop("_graphs")
    print("C2 step vs oracle:", rep)
    _sampler_bit_exact(g1, {"graph_q": q, "graph_k": k}, 0)


def test_moco_step_in_the_bench_configuration_vs_oracle():
    """the same check with the producer pipeline of bench.py switched on (2 lanes x depth 2 x chunks sampled by
    gcc_sample_multi on side streams, the step on the high-priority stream) and at step 3 of the run: the batch the
    checker reads back is the one the pipelined step consumed."""
    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from gcc_amd.train_step import MoCoTrainStep
    from tests.headline_step_check import check_moco_step

    g1 = _g1()
    torch.manual_seed(1)
    model, ema = _encoder(), _encoder()
    ema.load_state_dict(model.state_dict())
    contrast = MemoryMoCo(64, None, K, 0.07, use_softmax=True).cuda()
    chunk, depth = 4, 2
    lanes = []
    for _ in range(2):
        smp = DeviceRWRSampler(g1["graph"], B, run_seed=RUN_SEED, num_buffers=depth * chunk, max_steps=chunk)
        lanes.append((smp, DevicePosEmb(B, smp.node_cap, 32, device="cuda:0", seed=RUN_SEED, num_buffers=depth * chunk,
                                        max_views=2 * chunk)))
    tr = MoCoTrainStep(model, ema, contrast, lanes[0][0], lanes[0][1], lanes=lanes, depth=depth, chunk=chunk, prefetch=True)
    tr.dropout_seed = 11
    for i in range(3):                                   # three ordinary steps (in-kernel Philox dropout) first
        tr.step(i, 0.005)
    masks = (torch.rand(5, B, 64) >= 0.5).float().cuda().contiguous()
    rep = check_moco_step(tr, model, ema, contrast, 0.004, masks, sync=torch.cuda.synchronize, step_id=3)
    assert tr.check_status(strict_posemb=True) == 0
    assert rep["nodes_q"] > 10 * B
    q, k = rep.pop("_graphs")
    print("C2 pipelined step 3 vs oracle:", rep)
    _sampler_bit_exact(g1, {"graph_q": q, "graph_k": k}, 3 * B)


def test_e2e_step_at_bsz_256_vs_oracle():
    from gcc_amd.train_step import E2ETrainStep
    from tests.headline_step_check import check_e2e_step

    g1 = _g1()
    torch.manual_seed(2)
    model = _encoder()
    smp, pe = _lane(g1["graph"])
    tr = E2ETrainStep(model, smp, pe, prefetch=False)
    mq = (torch.rand(5, B, 64) >= 0.5).float().cuda().contiguous()
    mk = (torch.rand(5, B, 64) >= 0.5).float().cuda().contiguous()
    rep = check_e2e_step(tr, model, 0.005, mq, mk, sync=torch.cuda.synchronize)
    assert tr.check_status(strict_posemb=True) == 0
    assert rep["nodes_q"] > 10 * B
    print("E2E bsz 256 vs oracle:", rep)
