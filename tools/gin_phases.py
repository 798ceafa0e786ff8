"""Per-phase wall-clock ticks of gin_in_kernel and gin_mid_kernel (training stream alone, placeholder positional embedding): mean over
the workgroups that had a tile, us per tile; every workgroup times itself into its own slots."""
import sys
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gcc_amd import _cabi
from gcc_amd.contrast import MemoryMoCo
from gcc_amd.encoder import GraphEncoder
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.posemb import PlaceholderPosEmb
from gcc_amd.sampler import DeviceRWRSampler
from gcc_amd.train_step import MoCoTrainStep

dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
sampler = DeviceRWRSampler(graph, 256, run_seed=0, num_buffers=3)
enc_kw = dict(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512, freq_embedding_size=16,
              degree_embedding_size=16, output_dim=64, node_hidden_dim=64, edge_hidden_dim=64, num_layers=5,
              num_step_set2set=6, num_layer_set2set=3, norm=True, gnn_model="gin", degree_input=True)
torch.manual_seed(0)
model, ema = GraphEncoder(**enc_kw).to(dev), GraphEncoder(**enc_kw).to(dev)
ema.load_state_dict(model.state_dict())
contrast = MemoryMoCo(64, None, 16384, 0.07, use_softmax=True).to(dev)
ph = PlaceholderPosEmb(sampler.node_cap, 32, device=dev)
trainer = MoCoTrainStep(model, ema, contrast, sampler, ph, lanes=[(sampler, ph)], depth=2, graph=False)   # eager: a replayed graph carries the NULL ticks pointer it was captured with
for i in range(20):
    trainer.step(i, 0.005)
torch.cuda.synchronize()
lib = _cabi.load()
ticks = torch.zeros(3 * 16 * 2048, dtype=torch.int64, device=dev)
lib.gcc_gin_debug_ticks(ticks.data_ptr())
n = 20
for i in range(n):
    trainer.step(20 + i, 0.005)
torch.cuda.synchronize()
lib.gcc_gin_debug_ticks(None)
t = ticks.cpu().numpy().reshape(3, 16, 2048).astype(float)
names = {0: ["bn-table", "own-rows", "pool", "gather", "agg-store", "mfma+epilogue", "flush-stats"],
         2: ["node-count", "statistics+table", "weights->LDS", "rows+normalise", "products+stores", "barrier", "flush-stats"]}
names[1] = names[0]
for kind, nm in enumerate(["gin_in first layer", "gin_in other layers", "gin_mid"]):
    tiles = t[kind, 15]
    live = tiles > 0                                   # workgroups that had a tile
    per = t[kind, :7][:, live] / tiles[live] / 100.0   # us per tile and workgroup (100 MHz clock)
    print(f"{nm:20s} workgroups {int(live.sum()):5d} | " + "  ".join(f"{x} {per[i].mean():6.2f}" for i, x in enumerate(names[kind]))
          + f" | sum {per.sum(axis=0).mean():6.2f} us (slowest workgroup {per.sum(axis=0).max():6.2f})")
