"""Find the subgraphs whose positional embedding sets a status flag (bit 8: an eigenvector could not be produced / a
restart cap was hit) on the bench workload, one view and then one subgraph at a time, and dump them for CPU analysis
(gpurun_out/flagged_*.npz: local CSR of the subgraph).

    python tools/posemb_find_flagged.py [--steps 16] [--first 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.posemb import DevicePosEmb
from gcc_amd.sampler import BatchedCSR, DeviceRWRSampler

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--first", type=int, default=0)
ap.add_argument("--out", default="gpurun_out")
a = ap.parse_args()
dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
B = 256
sampler = DeviceRWRSampler(graph, B, run_seed=0, num_buffers=2)
pe = DevicePosEmb(B, sampler.node_cap, 32, device=dev, seed=0, num_buffers=2, max_views=1)
pe1 = DevicePosEmb(1, sampler.node_cap, 32, device=dev, seed=0, num_buffers=2, max_views=1)
i32 = dict(dtype=torch.int32, device=dev)
found = 0
for step in range(a.first, a.first + a.steps):
    for vi, view in enumerate(sampler.sample(step * B)):
        pe.status.zero_()
        pe(view)
        st = pe.status.cpu().tolist()
        if not st[0]:
            continue
        c = view.csr_numpy()
        no, rptr, col = c["node_off"], c["row_ptr"], c["col_idx"]
        for b in range(B):
            lo, hi = int(no[b]), int(no[b + 1])
            n = hi - lo
            lrp = (rptr[lo:hi + 1] - rptr[lo]).astype(np.int32)
            lci = (col[rptr[lo]:rptr[hi]] - lo).astype(np.int32)
            q = BatchedCSR(1, torch.tensor([0, n], **i32), torch.tensor([0, len(lci)], **i32), torch.zeros(n, **i32),
                           torch.zeros(n, **i32), torch.from_numpy(lrp).to(dev), torch.from_numpy(lci).to(dev))
            pe1.status.zero_()
            pe1(q)
            s1 = pe1.status.cpu().tolist()
            if s1[0]:
                deg = np.diff(lrp)
                print(f"step {step} view {vi} subgraph {b}: n={n} nnz={len(lci)} status={s1} leaves={(deg == 1).sum()}")
                np.savez(os.path.join(a.out, f"flagged_s{step}_v{vi}_b{b}.npz"), row_ptr=lrp, col_idx=lci)
                found += 1
        print(f"step {step} view {vi}: batch status {st}")
print("flagged subgraphs:", found)
