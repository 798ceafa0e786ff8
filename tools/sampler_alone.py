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
a = ap.parse_args()
dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(a.nodes, a.edges, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=a.rw_hops, restart_prob=0.8, device=dev, validate=False)
S = a.steps_per_call
sampler = DeviceRWRSampler(graph, a.batch_size, run_seed=0, num_buffers=max(2, S), max_steps=S)
for i in range(a.launches):
    if S > 1:
        sampler.sample_multi(10_000_000 + i * S * a.batch_size, S)
    else:
        sampler.sample(10_000_000 + i * a.batch_size)
    torch.cuda.synchronize()
sampler.check_status()
print("ok workload %d/%d/bsz%d/hops%d%s launches %d" % (len(rp) - 1, len(ci), a.batch_size, a.rw_hops,
                                                        "/steps%d" % S if S > 1 else "", a.launches))
