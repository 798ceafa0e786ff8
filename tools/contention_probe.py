"""Which kind of background work slows the training stream?  Foreground: the training step with a placeholder
positional embedding (1 producer lane = sampler only).  Background: one stream looping the positional-embedding
kernels on a synthetic batch whose subgraphs all fall into ONE solver class.  Prints ms/step per background kind."""
import sys, time
import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gcc_amd.contrast import MemoryMoCo
from gcc_amd.encoder import GraphEncoder
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.posemb import DevicePosEmb, PlaceholderPosEmb
from gcc_amd.sampler import BatchedCSR, DeviceRWRSampler
from gcc_amd.train_step import MoCoTrainStep

dev = torch.device("cuda:0")


def leafless(n, seed):
    rng = np.random.RandomState(seed)
    w = 1.0 / np.arange(1, n + 1) ** 0.5
    pr = np.minimum(1.0, 6.0 * np.outer(w, w) / w.mean())
    up = np.triu(rng.rand(n, n) < pr, 1)
    up[np.arange(n - 1), np.arange(1, n)] = True
    a = sp.csr_matrix((up | up.T).astype(np.float64))
    a.sort_indices()
    return a


def batch_of(n, copies):
    a = sp.block_diag([leafless(n, i) for i in range(copies)], format="csr")
    a.sort_indices()
    i32 = dict(dtype=torch.int32, device=dev)
    N = n * copies
    q = BatchedCSR(copies, torch.arange(0, N + 1, n, **i32), torch.zeros(copies + 1, **i32), torch.zeros(N, **i32),
                   torch.zeros(N, **i32), torch.from_numpy(a.indptr.astype(np.int32)).to(dev),
                   torch.from_numpy(a.indices.astype(np.int32)).to(dev))
    return q, DevicePosEmb(copies, N, 32, device=dev, seed=1)


rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
B = 256
sampler = DeviceRWRSampler(graph, B, run_seed=0, num_buffers=3)
enc_kw = dict(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512, freq_embedding_size=16,
              degree_embedding_size=16, output_dim=64, node_hidden_dim=64, edge_hidden_dim=64, num_layers=5,
              num_step_set2set=6, num_layer_set2set=3, norm=True, gnn_model="gin", degree_input=True)
torch.manual_seed(0)
model, ema = GraphEncoder(**enc_kw).to(dev), GraphEncoder(**enc_kw).to(dev)
ema.load_state_dict(model.state_dict())
contrast = MemoryMoCo(64, None, 16384, 0.07, use_softmax=True).to(dev)
ph = PlaceholderPosEmb(sampler.node_cap, 32, device=dev)
trainer = MoCoTrainStep(model, ema, contrast, sampler, ph, lanes=[(sampler, ph, ph)], depth=2)
for i in range(20):
    trainer.step(i, 0.005)
torch.cuda.synchronize()

import os
spec_by_name = {"small": (48, 64), "mid": (112, 48), "slot2": (300, 2), "slot14": (300, 14), "kry1": (700, 1), "kry6": (700, 6)}
step = 20
cases = [("none", 0)] + [(k, ns) for k in os.environ.get("PROBE_KINDS", "slot2,kry1,small").split(",") for ns in [int(x) for x in os.environ.get("PROBE_STREAMS", "1,2,4,8,16").split(",")]]
streams = [torch.cuda.Stream(dev) for _ in range(16)]
for name, nstreams in cases:
    pes = []
    if nstreams:
        for i in range(nstreams):
            pes.append(batch_of(*spec_by_name[name]))
        q, pe = pes[0]
        with torch.cuda.stream(streams[0]):
            pe(q); torch.cuda.synchronize()
            t = time.perf_counter(); pe(q); streams[0].synchronize(); one = time.perf_counter() - t
        reps = max(2, int(0.2 / max(one, 1e-4)))
        for i, (q, pe) in enumerate(pes):
            with torch.cuda.stream(streams[i]):
                for _ in range(reps):
                    pe(q)
    t0 = time.perf_counter()
    n = 60
    for _ in range(n):
        trainer.step(step, 0.005); step += 1
    trainer.main.synchronize()
    fg = (time.perf_counter() - t0) / n * 1e3
    still = not all(st.query() for st in streams)
    torch.cuda.synchronize()
    print(f"{name:8s} bg streams {nstreams:2d}  fg {fg:6.3f} ms/step  (bg still running at end: {still})", flush=True)
