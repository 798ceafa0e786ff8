"""The sampler of the default bench workload on an otherwise idle GPU (for rocprofv3 --kernel-trace --stats: the
per-kernel averages that bench.py's `kernel_ms_isolated` / `roofline` are compared with)."""
import sys
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.sampler import DeviceRWRSampler

dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False)
sampler = DeviceRWRSampler(graph, 256, run_seed=0, num_buffers=2)
for i in range(110):
    sampler.sample(10_000_000 + i * 256)
    torch.cuda.synchronize()
sampler.check_status()
print("ok")
