"""How much of the training stream's time is launch overhead?  The ~50 kernels of one MoCo step (encoder q+k forward,
head, backward, clip + Adam, EMA) on a FIXED sampled batch, GPU otherwise idle: N eager steps against N replays of the
same step captured in a hipGraph (torch.cuda.CUDAGraph: stream capture of the C-ABI launches).  An upper bound of what
graph capture of the step could buy (a real step changes buffers and scalars every time).

    python tools/graph_probe.py [--steps 200]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.contrast import MemoryMoCo
from gcc_amd.encoder import GraphEncoder
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.posemb import PlaceholderPosEmb
from gcc_amd.sampler import DeviceRWRSampler
from gcc_amd.train_step import MoCoTrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--strict-streams", action="store_true", help="keep the per-step caller <-> step stream hand-offs")
ap.add_argument("--lib", default=None, help="an ablation build of the library (timing experiments, e.g. -DGIN_DBG_SKIP=1)")
a = ap.parse_args()
if a.lib:
    import ctypes

    from gcc_amd import _cabi
    _cabi._lib = _cabi.declare(ctypes.CDLL(os.path.abspath(a.lib)))
dev = torch.device("cuda:0")
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


tr = MoCoTrainStep(model, ema, contrast, Fixed(), posemb=lambda g, prof=None: g, prefetch=False)
for i in range(20):
    tr.step(i, 0.005)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    tr.step(20 + i, 0.005)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / a.steps * 1e3

# host cost of issuing a step: a burst short enough for the hardware queue to take without back-pressure, timed WITHOUT a
# synchronisation (if this is close to the step time, the host thread -- which also issues the producer lanes' launches in
# the real pipeline -- is what the step waits for)
burst = 24
t0 = time.perf_counter()
for i in range(burst):
    tr.step(20 + a.steps + i, 0.005)
host = (time.perf_counter() - t0) / burst * 1e3
torch.cuda.synchronize()
print(f"host time to issue one step (Python + {burst}-step burst, no synchronisation): {host:.3f} ms")

# the SHIPPED graph path (MoCoTrainStep(graph=True): per-slot capture, device-resident scalars written by a one-thread launch
# in front of every replay): host time to issue a step, and the stream's time per step
tg = MoCoTrainStep(model, ema, contrast, Fixed(), posemb=lambda g, prof=None: g, prefetch=False, graph=True)
tg.relaxed_streams = "--strict-streams" not in sys.argv
for i in range(4):
    tg.step(30000 + i, 0.005)
torch.cuda.synchronize()
assert tg.graph_replays >= 2, tg.graph_replays
t0 = time.perf_counter()
for i in range(burst):
    tg.step(31000 + i, 0.005)
host_g = (time.perf_counter() - t0) / burst * 1e3
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    tg.step(32000 + i, 0.005)
torch.cuda.synchronize()
shipped = (time.perf_counter() - t0) / a.steps * 1e3
print(f"shipped graph path (MoCoTrainStep graph=True): host time to issue one step {host_g:.3f} ms; {shipped:.3f} ms/step on the stream")

side = torch.cuda.Stream(dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    tr.step(100000, 0.005)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        tr.step(100001, 0.005)
torch.cuda.synchronize()
for i in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    g.replay()
torch.cuda.synchronize()
graphed = (time.perf_counter() - t0) / a.steps * 1e3
print(f"training stream alone, fixed batch: eager {eager:.3f} ms/step, hipGraph replay {graphed:.3f} ms/step "
      f"({100.0 * (1.0 - graphed / eager):.1f} % less)")
