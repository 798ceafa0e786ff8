"""The pipelined producer at bench.py's shape (3 lanes x depth 2 x chunks of 10 steps, real sampler and eigensolver
rings on side streams, the step on the high-priority stream) against the SAME 40 steps produced sequentially on the
training stream (one lane, one chunk in flight, no look-ahead): a ring slot that is overwritten before its consumer has
read it, or read before its producer has finished, shows as a step whose subgraphs / positional embedding / loss
differ.  (Two pipelined runs compared with each other -- tests/test_rccl_gpu.py -- would not see a systematic hazard.)

Reference: train.py:378-434 (the step), train.py:577-586 (the DataLoader prefetch this replaces)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, CHUNK, DEPTH, LANES, STEPS = 256, 10, 2, 3, 40


def _digest(g):
    """Device-side digest of one batch view (live extent only: the ring buffers keep stale tails)."""
    n = g.node_off[g.batch_size].long()
    e = g.edge_off[g.batch_size].long()
    cap, ecap = g.parent_nid.numel(), g.col_idx.numel()
    ar = torch.arange(cap, device=g.parent_nid.device)
    live = (ar < n)
    w = (ar % 1000003 + 1)
    nid = (g.parent_nid.long() * w * live).sum()
    ae = torch.arange(ecap, device=g.col_idx.device)
    col = (g.col_idx.long() * (ae % 1000003 + 1) * (ae < e)).sum()
    pos = g.pos_undirected[:cap] * live[:, None]
    norms = pos.norm(dim=1)
    unit_err = torch.where(norms > 0, (norms - 1).abs(), torch.zeros_like(norms)).max()
    psum = (pos.double() * (w % 97 + 1)[:, None].double()).sum()
    pabs = pos.double().abs().sum()
    return torch.stack([n, e, nid, col]), torch.stack([psum, pabs, unit_err.double()])


def _run(pipelined, graph, steps=STEPS, stress=False):
    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.encoder import GraphEncoder
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from gcc_amd.train_step import BatchProducer, MoCoTrainStep

    dev = graph.device
    torch.manual_seed(0)
    kw = dict(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512, freq_embedding_size=16,
              degree_embedding_size=16, output_dim=64, node_hidden_dim=64, edge_hidden_dim=64, num_layers=5,
              num_step_set2set=6, num_layer_set2set=3, norm=True, gnn_model="gin", degree_input=True)
    model, ema = GraphEncoder(**kw).to(dev), GraphEncoder(**kw).to(dev)
    ema.load_state_dict(model.state_dict())
    contrast = MemoryMoCo(64, None, 16384, 0.07, use_softmax=True).to(dev)
    nl = LANES if pipelined else 1
    nbuf = (DEPTH if pipelined else 1) * CHUNK
    lanes = []
    for _ in range(nl):
        # pipelined: a chunk's batches in one gcc_sample_multi launch set; sequential: one gcc_sample_batch per step
        smp = DeviceRWRSampler(graph, B, run_seed=0, num_buffers=nbuf, max_steps=CHUNK if pipelined else 1)
        lanes.append((smp, DevicePosEmb(B, smp.node_cap, 32, device=dev, seed=0, num_buffers=nbuf, max_views=2 * CHUNK)))
    tr = MoCoTrainStep(model, ema, contrast, lanes[0][0], lanes[0][1], lanes=lanes, depth=DEPTH, chunk=CHUNK,
                       prefetch=pipelined)
    if not pipelined:
        # sequential: the same chunking (one multi-view eigensolver call per 10 steps, so every item keeps its id and
        # its start vectors), produced on the training stream right before its first step is consumed
        tr.producer = BatchProducer(lanes, tr._first_id, "cpu", depth=1, chunk=CHUNK)
        tr.producer.cuda = False
    tr.dropout_seed = 7
    losses, ints, flts = [], [], []
    if stress:
        # the hazard the round-5 fix (ad9f506) was about, provoked on purpose: no per-step synchronisation, no caller <-> step stream
        # hand-offs (relaxed_streams), graph replay, and the producer lanes held up by random device-side sleeps so that refills,
        # look-ahead launches and the consumer interleave differently at every step.  The batches are read ON THE STEP'S STREAM (the
        # documented way, MoCoTrainStep.step): a slot handed back too early or consumed too early shows as a differing digest.
        import random

        rnd = random.Random(1234)
        tr.relaxed_streams = True
        for i in range(steps):
            if rnd.random() < 0.35:
                lane_stream = rnd.choice(tr.producer.streams)
                with torch.cuda.stream(lane_stream):
                    torch.cuda._sleep(int(rnd.uniform(2e5, 4e6)))          # 0.1 - 2 ms at ~2 GHz
            if rnd.random() < 0.15:
                with torch.cuda.stream(tr.main):
                    torch.cuda._sleep(int(rnd.uniform(2e5, 2e6)))
            out = tr.step(i, 0.005)
            with torch.cuda.stream(tr.main):
                losses.append(out["loss"].reshape(()).clone())
                dq, dk = _digest(out["graph_q"]), _digest(out["graph_k"])
                ints.append(torch.stack([dq[0], dk[0]]))
                flts.append(torch.stack([dq[1], dk[1]]))
        torch.cuda.synchronize()
        assert tr.check_status(strict_posemb=True) == 0
        assert getattr(tr, "graph_capture_failures", 0) == 0 and tr.graph_replays > steps // 2
        return torch.stack(losses).cpu(), torch.stack(ints).cpu(), torch.stack(flts).cpu()
    for i in range(steps):
        out = tr.step(i, 0.005)
        losses.append(out["loss"].reshape(()).clone())
        dq, dk = _digest(out["graph_q"]), _digest(out["graph_k"])
        ints.append(torch.stack([dq[0], dk[0]]))
        flts.append(torch.stack([dq[1], dk[1]]))
        # (no synchronisation here: the digests run on THIS stream, the step on the trainer's; the next step() waits for this stream
        #  and only then hands the consumed ring slots back -- round 5 recorded the "slot free" event before these reads and needed
        #  a synchronisation per step here)
    torch.cuda.synchronize()
    flags = tr.check_status(strict_posemb=True)
    assert flags == 0
    return torch.stack(losses).cpu(), torch.stack(ints).cpu(), torch.stack(flts).cpu()


def test_pipelined_producer_equals_sequential_production_step_by_step():
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph

    dev = torch.device("cuda:0")
    rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
    graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
    seq_loss, seq_i, seq_f = _run(False, graph)
    pip_loss, pip_i, pip_f = _run(True, graph)
    assert int(seq_i[:, :, 0].min()) > B                   # live batches, not empty buffers
    # sampler outputs: integer work, bit-exact (step, view, [nodes, edges, parent_nid digest, col_idx digest])
    bad = (seq_i != pip_i).nonzero()
    assert bad.numel() == 0, bad[:8].tolist()
    # positional embedding: the same deterministic solver on the same items
    torch.testing.assert_close(pip_f[:, :, 0], seq_f[:, :, 0], rtol=1e-6, atol=1e-3)
    torch.testing.assert_close(pip_f[:, :, 1], seq_f[:, :, 1], rtol=1e-6, atol=1e-3)
    assert float(pip_f[:, :, 2].max()) < 1e-4 and float(seq_f[:, :, 2].max()) < 1e-4   # unit rows (data_util.py:260)
    # and the step
    assert torch.isfinite(seq_loss).all()
    torch.testing.assert_close(pip_loss, seq_loss, rtol=2e-5, atol=1e-6)


def test_ring_slots_under_random_producer_delays():
    """120 steps of the pipelined producer with random device-side delays on the lanes' and the step's streams, relaxed stream
    hand-offs and graph replay, read on the step's stream without a host synchronisation in the loop, against sequential
    production: every step's sampler digests bit-equal, positional-embedding digests and losses equal."""
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph

    dev = torch.device("cuda:0")
    rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
    graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
    n = 120
    seq_loss, seq_i, seq_f = _run(False, graph, steps=n)
    pip_loss, pip_i, pip_f = _run(True, graph, steps=n, stress=True)
    bad = (seq_i != pip_i).nonzero()
    assert bad.numel() == 0, bad[:8].tolist()
    torch.testing.assert_close(pip_f[:, :, 0], seq_f[:, :, 0], rtol=1e-6, atol=1e-3)
    torch.testing.assert_close(pip_f[:, :, 1], seq_f[:, :, 1], rtol=1e-6, atol=1e-3)
    torch.testing.assert_close(pip_loss, seq_loss, rtol=2e-5, atol=1e-6)
