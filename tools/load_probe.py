"""What does the training stream lose next to a co-tenant?  The fused MoCo step on a FIXED batch (graph replay, as
tools/graph_probe.py), alone and while ONE long synthetic kernel (gcc_debug_load) holds a number of workgroups on another
stream: barriers + LDS only (occupancy: wave slots and LDS, no memory traffic, no arithmetic -- how a latency-bound solver
workgroup looks to its neighbours), float4 reads streamed over an L2-sized / an HBM-sized buffer (the memory system),
independent FMA chains (issue slots and power).  Which resource the eigensolvers take from the step decides what to
change in them.

    python tools/load_probe.py [--steps 120]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import _cabi
from gcc_amd.contrast import MemoryMoCo
from gcc_amd.encoder import GraphEncoder
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.posemb import PlaceholderPosEmb
from gcc_amd.sampler import DeviceRWRSampler
from gcc_amd.train_step import MoCoTrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=120)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _cabi.load()
rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
B = 256
torch.manual_seed(0)
sampler = DeviceRWRSampler(graph, B, run_seed=0, num_buffers=2)
kw = dict(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512, freq_embedding_size=16,
          degree_embedding_size=16, output_dim=64, node_hidden_dim=64, edge_hidden_dim=64, num_layers=5,
          num_step_set2set=6, num_layer_set2set=3, norm=True, gnn_model="gin", degree_input=True)
model, ema = GraphEncoder(**kw).to(dev), GraphEncoder(**kw).to(dev)
ema.load_state_dict(model.state_dict())
contrast = MemoryMoCo(64, None, 16384, 0.07, use_softmax=True).to(dev)
pe = PlaceholderPosEmb(sampler.node_cap, 32, device=dev)
q, k = sampler.sample(0)
pe(q); pe(k)


class Fixed:
    batch_size = B

    def sample(self, first_id, prof=None):
        return q, k


tr = MoCoTrainStep(model, ema, contrast, Fixed(), posemb=lambda g, prof=None: g, prefetch=False, graph=True)
tr.relaxed_streams = True
for i in range(8):
    tr.step(i, 0.005)
torch.cuda.synchronize()
side = torch.cuda.Stream(dev)
sink = torch.zeros(4, device=dev)
small = torch.randn(2 << 20, device=dev)            # 8 MB: stays in L2 / Infinity Cache
large = torch.randn(512 << 20, device=dev)          # 2 GB: HBM


done = [8]


def steps(n):
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(done[0], 0.005)
        done[0] += 1
    tr.join()
    torch.cuda.current_stream(dev).synchronize()
    return (time.perf_counter() - t0) / n * 1e3


alone = steps(a.steps)
print(f"training stream alone (fixed batch, graph replay): {alone:.3f} ms per step")
cases = [("barriers + LDS, 96 x 1024 threads, 140 KB", 0, 96, 1024, 140 << 10, None),
         ("barriers + LDS, 96 x 1024 threads, 130 KB (one 26 KB training workgroup fits beside it)", 0, 96, 1024, 130 << 10, None),
         ("barriers + LDS, 96 x 1024 threads, 104 KB (two fit)", 0, 96, 1024, 104 << 10, None),
         ("barriers + LDS, 96 x 1024 threads, 78 KB (three fit)", 0, 96, 1024, 78 << 10, None),
         ("barriers + LDS, 96 x 1024 threads, 52 KB (four fit)", 0, 96, 1024, 52 << 10, None),
         ("barriers + LDS, 96 x 512 threads, 140 KB", 0, 96, 512, 140 << 10, None),
         ("barriers + LDS, 160 x 1024 threads, 104 KB (two fit)", 0, 160, 1024, 104 << 10, None),
         ("barriers + LDS, 160 x 1024 threads, 78 KB (three fit)", 0, 160, 1024, 78 << 10, None),
         ("barriers + LDS, 160 x 1024 threads, 140 KB", 0, 160, 1024, 140 << 10, None),
         ("barriers + LDS, 96 x 1024 threads, 16 KB", 0, 96, 1024, 16 << 10, None),
         ("barriers + LDS, 256 x 256 threads, 16 KB", 0, 256, 256, 16 << 10, None),
         ("reads over 8 MB (L2 / Infinity Cache), 96 x 1024", 1, 96, 1024, 8 << 10, small),
         ("reads over 8 MB (L2 / Infinity Cache), 160 x 1024", 1, 160, 1024, 8 << 10, small),
         ("reads over 2 GB (HBM), 96 x 1024", 1, 96, 1024, 8 << 10, large),
         ("FMA chains, 96 x 1024", 2, 96, 1024, 8 << 10, None),
         ("FMA chains, 160 x 1024", 2, 160, 1024, 8 << 10, None)]
for name, kind, wgs, thr, lds, buf in cases:
    ticks = int((a.steps * alone * 1.8 + 20.0) * 1e5)          # 100 MHz: the co-tenant outlasts the timed steps
    with torch.cuda.stream(side):
        rc = lib.gcc_debug_load(kind, wgs, thr, lds, ticks, 1 << 30, buf.data_ptr() if buf is not None else None,
                                buf.numel() if buf is not None else 0, sink.data_ptr(), side.cuda_stream)
    assert rc == 0, lib.gcc_last_error()
    time.sleep(0.005)                                           # the co-tenant is resident before the first step
    t = steps(a.steps)
    still = not side.query()
    torch.cuda.synchronize()
    print(f"next to {name}: {t:.3f} ms per step ({t / alone:.2f}x){'' if still else '   [the co-tenant ended before the steps did]'}")
