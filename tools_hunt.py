import sys, numpy as np, torch
sys.path.insert(0, '.')
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.sampler import DeviceRWRSampler
rp, ci = powerlaw_graph(1_000_000, 10_000_000, 0)
g = DeviceGraph(rp, ci, rw_hops=256, validate=False)
s = DeviceRWRSampler(g, 256, run_seed=0)
for step in range(0, 400):
    q, k = s.sample(step * 256)
    torch.cuda.synchronize()
    st = int(s.status.item())
    nq, ek = q.number_of_nodes(), k.number_of_edges()
    print(step, st, nq, ek, flush=True)
