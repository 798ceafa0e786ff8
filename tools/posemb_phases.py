"""Per-phase wall-clock ticks of the positional-embedding solver classes on a real sampled chunk (isolated GPU)."""
import sys
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gcc_amd import _cabi
from gcc_amd.graph import DeviceGraph
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.posemb import DevicePosEmb
from gcc_amd.sampler import DeviceRWRSampler

dev = torch.device("cuda:0")
rp, ci = powerlaw_graph(1_000_000, 10_000_000, seed=0)
graph = DeviceGraph(rp, ci, rw_hops=256, restart_prob=0.8, device=dev, validate=False, trusted=True)
B, S = 256, 8
sampler = DeviceRWRSampler(graph, B, run_seed=0, num_buffers=S)
pe = DevicePosEmb(B, sampler.node_cap, 32, device=dev, seed=0, num_buffers=S, max_views=2 * S)
views = [g for s in range(S) for g in sampler.sample(10_000_000 + s * B)]
lib = _cabi.load()
NCLS = 8      # GCC_POSEMB_TICK_CLASSES
ticks = torch.zeros(NCLS * 16, dtype=torch.int64, device=dev)
pe.multi(views); torch.cuda.synchronize()
lib.gcc_posemb_debug_ticks(ticks.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); pe.multi(views); e1.record(); torch.cuda.synchronize()
lib.gcc_posemb_debug_ticks(None)
t = ticks.cpu().numpy().reshape(NCLS, 16)
print("multi call of %d views: %.2f ms" % (len(views), e0.elapsed_time(e1)))
names = {0: ["matrix", "tridiag", "bisect", "invit", "gram-schmidt", "backtransf", "expand"], 3: ["arnoldi", "ritz(H)", "restart", "final"],
         5: ["matrix", "sparse-products", "ritz", "rotate", "expand", "gram", "cholesky", "inverse+H", "c-matrix"]}
grand = 0.0
for c, cname in enumerate(["small", "mid", "slot", "krylov", "big", "cheb", "wave48", "wave64"]):
    items = max(int(t[c, 15]), 1)
    ph = names.get(c, names[0])
    share = 4.0 if cname.startswith("wave") else 1.0    # one-wave teams: 4 items in flight per workgroup, ticks are wave time
    if cname == "mid" and __import__("os").environ.get("GCC_POSEMB_PAIR", "1") != "0":
        share = float(__import__("os").environ.get("GCC_POSEMB_TEAM_SHARE", "3"))   # four-wave teams: three 256-thread workgroups share a CU (168 registers; 33 KiB of LDS) -- 4 for the two-wave build
    tot = t[c, :len(ph)].sum() / 100.0 / share          # us of workgroup residency
    grand += tot
    print(f"{cname:7s} items {items:5d}  total {tot/1e3:8.2f} CU-ms  per item {tot/items:8.1f} us  | " +
          "  ".join(f"{n} {t[c, i]/100.0/items:7.1f}" for i, n in enumerate(ph)))
print("total %.2f CU-s per call" % (grand / 1e6))
print("status", pe.status.cpu().tolist())
