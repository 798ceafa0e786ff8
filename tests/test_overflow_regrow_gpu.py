"""A sampler whose induction scratch / edge capacity is too small for its batches does not kill the run any more: the
producer reads the chunk's overflow flags before the chunk is consumed, enlarges what overflowed and samples the chunk
again (gcc_amd/train_step.py: BatchProducer._reproduce, gcc_amd/sampler.py: DeviceRWRSampler.grow).  Every batch must be
bit for bit what an amply sized sampler produces.  Reference semantics being protected: graph_dataset.py:94-130 /
data_util.py:218-239 never truncate a subgraph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, CHUNK, STEPS = 64, 4, 16


def _graph():
    from gcc_amd.graph import DeviceGraph
    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(100_000, 1_000_000, 2)
    return DeviceGraph(rp, ci, rw_hops=128, device="cuda:0")


def _batches(graph, pipelined, **small):
    from gcc_amd.posemb import PlaceholderPosEmb
    from gcc_amd.sampler import DeviceRWRSampler
    from gcc_amd.train_step import BatchProducer

    lanes = []
    for _ in range(2 if pipelined else 1):
        smp = DeviceRWRSampler(graph, B, run_seed=3, num_buffers=2 * CHUNK, max_steps=CHUNK, **small)
        lanes.append((smp, PlaceholderPosEmb(smp.node_cap, 32, device="cuda:0")))
    bp = BatchProducer(lanes, lambda step: step * B, "cuda:0" if pipelined else "cpu", depth=2, chunk=CHUNK)
    if not pipelined:
        bp.cuda = False
    out = []
    for s in range(STEPS):
        q, k = bp.get(s)
        torch.cuda.synchronize()
        out.append((q.csr_numpy(), k.csr_numpy()))
        bp.release(s)
    return out, bp, lanes


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("small", [dict(scratch_entries=4096), dict(edge_cap=4096), dict(scratch_entries=4096, edge_cap=4096)])
def test_overflow_is_regrown_and_the_batches_are_exact(pipelined, small):
    graph = _graph()
    ref, bp0, _ = _batches(graph, False)
    assert bp0.regrown == 0
    got, bp, lanes = _batches(graph, pipelined, **small)
    assert bp.regrown > 0 and sum(l[0].regrown for l in lanes) > 0
    for s, ((rq, rk), (gq, gk)) in enumerate(zip(ref, got)):
        for r, g in ((rq, gq), (rk, gk)):
            for key in ("node_off", "edge_off", "parent_nid", "graph_id", "row_ptr", "col_idx"):
                assert np.array_equal(r[key], g[key]), (s, key)
    for l in lanes:
        l[0].check_status()                      # nothing left set


def test_hub_only_batch_completes():
    """every seed of the batch is one of the graph's largest hubs (the sizing heuristic's worst case)."""
    from gcc_amd.sampler import DeviceRWRSampler
    from oracle import sampler as O

    from gcc_amd.graphgen import powerlaw_graph

    rp, ci = powerlaw_graph(100_000, 1_000_000, 2)
    graph = _graph()
    hubs = np.argsort(np.diff(rp))[-B:].astype(np.int32)
    smp = DeviceRWRSampler(graph, B, run_seed=3, scratch_entries=1 << 16)
    seeds = torch.from_numpy(hubs).cuda()
    for attempt in range(4):
        q, k = smp.sample(0, seeds=seeds)
        tok = smp.status_snapshot()
        smp.snapshot_sync()
        bits = smp.read_snapshot(tok)
        if not bits:
            break
        smp.grow(bits)
    assert bits == 0 and smp.regrown >= 1
    c = O.COracle()
    L = O.max_nodes_table(int(np.diff(rp).max()), 128, 0.8)[np.diff(rp)[hubs]]
    for view, gb in enumerate((q, k)):
        r = c.sample_batch(rp, ci, hubs, L, view, 3, 0, O.restart_threshold(0.8))
        g = gb.csr_numpy()
        for key in ("parent_nid", "row_ptr", "col_idx"):
            assert np.array_equal(g[key], r[key]), (view, key)
