"""The sampler of a bench workload on an otherwise idle GPU (for rocprofv3 --kernel-trace --stats and the --pmc passes:
the per-kernel averages that bench.py's `kernel_ms_isolated` / `roofline` are compared with).

    python tools/sampler_alone.py [--nodes V --edges E --batch-size B --rw-hops H --launches N]
"""
import argparse
import sys
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.sampler import DeviceRWRSampler

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--batch-size", type=int, default=256)
ap.add_argument("--rw-hops", type=int, default=256)
ap.add_argument("--launches", type=int, default=110)
ap.add_argument("--steps-per-call", type=int, default=1, help="consecutive batches per launch set (gcc_sample_multi)")
ap.add_argument("--hub-degree", type=int, default=0, help="0 = default (512), -1 = scan every row")
ap.add_argument("--max-hubs", type=int, default=0, help="0 = default (8); at most 32")
ap.add_argument("--sweep", default="", help="hub_degree:max_hubs,... -- wall clock of back-to-back launches for each pair (one process, one graph)")
ap.add_argument("--hub-stats", action="store_true", help="members of degree >= 256 / 1024 / 4096 per subgraph of one launch")
ap.add_argument("--time", action="store_true", help="also: wall clock of the launches issued back to back (one synchronisation at the end)")
ap.add_argument("--lib", default=None, help="a variant build of the library (tools/build_variant.sh)")
a = ap.parse_args()
if a.lib:
    import ctypes
    import os

    from gcc_amd import _cabi
    _cabi._lib = _cabi.declare(ctypes.CDLL(os.path.abspath(a.lib)))
dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(a.nodes, a.edges, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=a.rw_hops, restart_prob=0.8, device=dev, validate=False, trusted=True)
S = a.steps_per_call
if a.sweep:
    import time
    for item in a.sweep.split(","):
        hd, mh = (int(x) for x in item.split(":"))
        sm = DeviceRWRSampler(graph, a.batch_size, run_seed=0, num_buffers=max(2, S), max_steps=S, hub_degree=hd, max_hubs=mh)
        run = (lambda i: sm.sample_multi(30_000_000 + i * S * a.batch_size, S)) if S > 1 else (lambda i: sm.sample(30_000_000 + i * a.batch_size))
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        sm.check_status()
        t0 = time.perf_counter()
        for i in range(a.launches):
            run(3 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.launches * 1e3
        sm.check_status()
        print("hub_degree %d max_hubs %d: %.4f ms per launch of %d step(s), %.0f subgraphs/s" % (hd, mh, ms, S, 2 * a.batch_size * S / ms * 1e3), flush=True)
        del sm
        torch.cuda.empty_cache()
    sys.exit(0)
sampler = DeviceRWRSampler(graph, a.batch_size, run_seed=0, num_buffers=max(2, S), max_steps=S, hub_degree=a.hub_degree, max_hubs=a.max_hubs)
for i in range(a.launches):
    if S > 1:
        sampler.sample_multi(10_000_000 + i * S * a.batch_size, S)
    else:
        sampler.sample(10_000_000 + i * a.batch_size)
    torch.cuda.synchronize()
sampler.check_status()
if a.hub_stats:
    q, k = sampler.sample(40_000_000)
    torch.cuda.synchronize()
    rpd = graph.row_ptr if hasattr(graph, "row_ptr") else None
    rpt = torch.as_tensor(rp, device=dev)
    for name, gq in (("q", q), ("k", k)):
        nn = int(gq.node_off[a.batch_size])
        nid = gq.parent_nid[:nn].long()
        deg = rpt[nid + 1] - rpt[nid]
        gid = gq.graph_id[:nn].long()
        sizes = torch.bincount(gid, minlength=a.batch_size).float()
        line = "view %s: members per subgraph mean %.1f max %d" % (name, sizes.mean().item(), int(sizes.max()))
        for T in (256, 1024, 4096, 16384):
            c = torch.bincount(gid, weights=(deg >= T).float(), minlength=a.batch_size)
            qs = torch.quantile(c, torch.tensor([0.5, 0.9, 1.0], device=dev))
            line += "; deg>=%d: mean %.1f p50 %d p90 %d max %d" % (T, c.mean().item(), int(qs[0]), int(qs[1]), int(qs[2]))
        ent = deg.float()
        line += "; row entries: all %.0f, in rows >=256 %.0f, >=1024 %.0f, >=4096 %.0f" % (
            ent.sum().item(), ent[deg >= 256].sum().item(), ent[deg >= 1024].sum().item(), ent[deg >= 4096].sum().item())
        print(line)
if a.time:
    import time
    t0 = time.perf_counter()
    for i in range(a.launches):
        if S > 1:
            sampler.sample_multi(20_000_000 + i * S * a.batch_size, S)
        else:
            sampler.sample(20_000_000 + i * a.batch_size)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.launches * 1e3
    sampler.check_status()
    print("hub_degree %d max_hubs %d: %.4f ms per launch of %d step(s), %.0f subgraphs/s" % (a.hub_degree, a.max_hubs, ms, S, 2 * a.batch_size * S / ms * 1e3))
print("ok workload %d/%d/bsz%d/hops%d%s launches %d" % (len(rp) - 1, len(ci), a.batch_size, a.rw_hops,
                                                        "/steps%d" % S if S > 1 else "", a.launches))
