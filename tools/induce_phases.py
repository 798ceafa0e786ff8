"""Per-phase wall-clock ticks of induce_kernel on the default bench workload (isolated GPU)."""
import argparse
import sys
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gcc_amd import _cabi
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.sampler import DeviceRWRSampler

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--edges", type=int, default=10_000_000)
ap.add_argument("--steps-per-call", type=int, default=1)
ap.add_argument("--hub-degree", type=int, default=0)
args = ap.parse_args()
S = args.steps_per_call
dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(args.nodes, args.edges, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
sampler = DeviceRWRSampler(graph, 256, run_seed=0, num_buffers=max(2, S), max_steps=S, hub_degree=args.hub_degree)
run = (lambda i: sampler.sample_multi(10_000_000 + i * 256 * S, S)) if S > 1 else (lambda i: sampler.sample(10_000_000 + i * 256))
for i in range(5):
    run(i)
torch.cuda.synchronize()
lib = _cabi.load()
ticks = torch.zeros(16, dtype=torch.int64, device=dev)
lib.gcc_sampler_debug_ticks(ticks.data_ptr())
n = 20
for i in range(n):
    run(5 + i)
torch.cuda.synchronize()
lib.gcc_sampler_debug_ticks(None)
t = ticks.cpu().numpy()
wgs = max(int(t[15]), 1)
print(f"hub_degree {args.hub_degree} steps {S}: workgroups per launch {wgs / n:.0f}; per workgroup: virtual-workgroup prefix copy {t[0] / 100 / wgs:.2f} us, member tables + Bloom "
      f"bitmap {t[1] / 100 / wgs:.2f} us, unit scans {t[2] / 100 / wgs:.2f} us")
