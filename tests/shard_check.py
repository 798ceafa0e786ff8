"""Shared pieces of the per-worker-shard sampling tests (graph_dataset.py:23-30,63-92).
TEST INFRASTRUCTURE ONLY."""
import numpy as np

from gcc_amd.graphgen import powerlaw_graph, tiny_graphs
from oracle import sampler as O


def corpus():
    """Graphs of very different size, as in small.bin: LPT over 2 workers puts the big one alone in shard 0."""
    return [powerlaw_graph(900, 7000, 4), powerlaw_graph(6000, 60000, 1), tiny_graphs()["k4"], powerlaw_graph(2500, 20000, 2)]


def reference_layout(graphs, num_workers, num_copies=1):
    """jobs of graph_dataset.py:63-76 restated independently of gcc_amd.sampler, the union laid out in jobs order,
    and the node ranges of the worker shards."""
    sizes = [len(rp) - 1 for rp, _ in graphs]
    bins = num_workers // num_copies
    jobs, load = [[] for _ in range(bins)], [0] * bins
    for idx, size in sorted(enumerate(sizes), key=lambda t: t[1], reverse=True):
        w = load.index(min(load))
        load[w] += size
        jobs[w].append(idx)
    order = [i for j in jobs for i in j]
    rps, cis, off, eoff = [np.zeros(1, np.int64)], [], 0, 0
    for i in order:
        rp, ci = graphs[i]
        rps.append(np.asarray(rp[1:], np.int64) + eoff)
        cis.append(np.asarray(ci, np.int64) + off)
        off += len(rp) - 1
        eoff += int(rp[-1])
    shard_off = np.cumsum([0] + [sum(sizes[i] for i in j) for j in jobs]).astype(np.int64)
    return jobs * num_copies, np.concatenate(rps).astype(np.int32), np.concatenate(cis).astype(np.int32), shard_off


def oracle_batch(coracle, rp, ci, shard_off, ltab, restart_u32, B, run_seed, first):
    """Seeds of batch `first // B` drawn by the C oracle out of that batch's worker shard, both views sampled."""
    cdf = O.seed_cdf(rp, shard_off)
    seeds = coracle.draw_seeds(cdf, run_seed, first, B, shard_off=shard_off, batch_size=B)
    # the independent Python restatement agrees, and every seed lies in the batch's shard
    sh = (first // B) % (len(shard_off) - 1)
    for b in range(B):
        assert O.py_draw_seed(cdf, run_seed, first + b, shard_off, B) == int(seeds[b])
    assert np.all((seeds >= shard_off[sh]) & (seeds < shard_off[sh + 1]))
    L = ltab[np.minimum(np.diff(rp)[seeds], len(ltab) - 1)]
    views = [coracle.sample_batch(rp, ci, seeds, L, v, run_seed, first, restart_u32) for v in range(2)]
    return seeds, views
