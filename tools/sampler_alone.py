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
a = ap.parse_args()
dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(a.nodes, a.edges, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=a.rw_hops, restart_prob=0.8, device=dev, validate=False)
sampler = DeviceRWRSampler(graph, a.batch_size, run_seed=0, num_buffers=2)
for i in range(a.launches):
    sampler.sample(10_000_000 + i * a.batch_size)
    torch.cuda.synchronize()
sampler.check_status()
print("ok workload %d/%d/bsz%d/hops%d launches %d" % (len(rp) - 1, len(ci), a.batch_size, a.rw_hops, a.launches))
