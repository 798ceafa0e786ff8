"""generate.py's embedding loop (generate.py:33-53) on a 10k-node graph: the eval-mode encoder as the 15-launch chain per
view + a torch mean, against gcc_gin_eval_fused (both views and the mean in ONE launch, one workgroup per subgraph).
Sampler and positional embedding are produced once per batch outside the timed loops (they are the same for both).

    python tools/eval_probe.py [--nodes 10000] [--edges 100000] [--batch-size 256] [--rw-hops 64] [--reps 50]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.encoder import GraphEncoder
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.posemb import DevicePosEmb
from gcc_amd.sampler import DeviceRWRSampler

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10000)
ap.add_argument("--edges", type=int, default=100000)
ap.add_argument("--batch-size", type=int, default=256)
ap.add_argument("--rw-hops", type=int, default=64)
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(a.nodes, a.edges, seed=1)
graph = DeviceGraph(rp, ci, rw_hops=a.rw_hops, device=dev)
B = a.batch_size
smp = DeviceRWRSampler(graph, B, run_seed=0)
pe = DevicePosEmb(B, smp.node_cap, 32, device=dev, seed=0, max_views=2)
q, k = smp.sample(0)
pe.multi([q, k])
smp.check_status()
torch.manual_seed(0)
model = GraphEncoder(positional_embedding_size=32, max_node_freq=16, max_edge_freq=16, max_degree=512,
                     freq_embedding_size=16, degree_embedding_size=16, output_dim=64, node_hidden_dim=64,
                     edge_hidden_dim=64, num_layers=5, num_step_set2set=6, num_layer_set2set=3, norm=True,
                     gnn_model="gin", degree_input=True).to(dev)
model.eval()
sizes = torch.diff(q.node_off[: B + 1]).cpu()


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


def chain():
    model.fused_eval = False
    with torch.no_grad():
        return (model(q) + model(k)) / 2


def fused():
    model.fused_eval = True
    with torch.no_grad():
        return model.embed_views(q, k)


tc, tf = timed(chain), timed(fused)
err = float((chain() - fused()).abs().max())

# the same two paths with the pass descriptors built once (the Python side -- ~170 pointers per pass -- is most of a call
# on this small model): the launches only, i.e. the GPU's time per batch
eng = model.engine()
st = torch.cuda.current_stream(dev).cuda_stream
pq, bq = eng.make_pass(model, q, training=False, slot=("probe", 0))
pk, bk = eng.make_pass(model, k, training=False, slot=("probe", 1))
mean = torch.zeros(B, 64, device=dev)


def chain_launches():
    eng.forward([pq], stream=st)
    eng.forward([pk], stream=st)
    torch.add(bq["feat"], bk["feat"], out=mean)
    mean.mul_(0.5)


def fused_launch():
    eng.eval_fused([pq, pk], mean_out=mean, stream=st)


gc, gf = timed(chain_launches), timed(fused_launch)


def graphed(fn):
    """the GPU's own time: the launches captured once in a hipGraph and replayed (no Python, no launch calls in the loop)"""
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.synchronize()
    return timed(g.replay)


st_keep = st
try:
    st = None          # (capture runs on the capturing stream: the closures read `st` at call time)
    def chain_cap():
        s_ = torch.cuda.current_stream(dev).cuda_stream
        eng.forward([pq], stream=s_)
        eng.forward([pk], stream=s_)
        torch.add(bq["feat"], bk["feat"], out=mean)
        mean.mul_(0.5)

    def fused_cap():
        eng.eval_fused([pq, pk], mean_out=mean, stream=torch.cuda.current_stream(dev).cuda_stream)

    hc, hf = graphed(chain_cap), graphed(fused_cap)
finally:
    st = st_keep
ticks = torch.zeros(32, dtype=torch.int64, device=dev)
ticks[8] = 2 ** 62          # (first start: a minimum)
eng.lib.gcc_gin_eval_debug_ticks(ticks.data_ptr())
fused_launch()
torch.cuda.synchronize()
eng.lib.gcc_gin_eval_debug_ticks(None)
tk = ticks.cpu().tolist()
names = ["features", "pooling", "weights", "own-rows", "gather", "linears", "mirror", "readout"]
med_names = ["features", "pooling", "weights", "aggregation(wave 0)", "products(wave 0)", "write-back", "wait for the other waves", "readout"]
phases = ""
for k, kern in enumerate(["LDS-resident kernel (subgraphs / runs of <= 320 nodes)", "general kernel"]):  # (every 4th group of four reports)
    wg = tk[16 * k + 15]
    if wg:
        phases += f"\n   {kern}: {wg} workgroups, us per workgroup: " + "  ".join(
            f"{n} {tk[16 * k + i] / 100.0 / wg:.1f}" for i, n in enumerate(med_names if k == 0 else names) if i < 8 and tk[16 * k + i])
        if k == 0 and tk[10]:
            phases += (f"\n      workgroups with work: last start {(tk[9] - tk[8]) / 100.0:.1f} us after the first, last end "
                       f"{(tk[10] - tk[8]) / 100.0:.1f} us after the first start, longest stay {tk[11] / 100.0:.1f} us")
print(f"graph {len(rp) - 1} nodes / {len(ci)} edges, batch {B} x 2 views, rw_hops {a.rw_hops}: subgraph sizes "
      f"median {int(sizes.median())} max {int(sizes.max())}; eval chain (2 x 15 launches + mean) {tc * 1e3:.1f} us per batch, "
      f"gcc_gin_eval_fused (1 launch) {tf * 1e3:.1f} us per batch = {tc / tf:.1f}x; max |difference| {err:.2e}; "
      f"launches only (descriptors built once): chain {gc * 1e3:.1f} us, fused {gf * 1e3:.1f} us = {gc / gf:.1f}x; "
      f"GPU time (hipGraph replay of the same launches): chain {hc * 1e3:.1f} us, fused {hf * 1e3:.1f} us = {hc / hf:.1f}x\n"
      f"   fused call, phase ticks per kernel:{phases}")
