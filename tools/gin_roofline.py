"""BASELINE.json configs[4]: "GIN hid=256 layers=8 deg=32 bf16, SpMM+MFMA-MLP roofline run on batched subgraphs,
1xMI355X".  4096 subgraphs x 128 nodes, exactly 32 random in-block in-neighbours per node (seed 0), features
~N(0,1) bf16.  Times gcc_ginw_forward two ways and prints one JSON line:
  fused      one launch, 8 layers, the subgraph stays in LDS          -> bound by the matrix cores
  layerwise  8 launches of one layer, rows through HBM in between    -> the per-layer SpMM + MLP pair of the config
FLOPs are algorithmic: per layer 2*nnz*256 (aggregation) + 2*N*(256*256*2) (the two Linear layers)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from gcc_amd.gin_wide import FoldedWideGIN

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subgraphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=128)
    ap.add_argument("--deg", type=int, default=32)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--phases", action="store_true", help="also report per-phase wall-clock of the fused launch")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, n, deg, L, D = args.subgraphs, args.nodes, args.deg, args.layers, 256
    N, nnz = B * n, B * n * deg
    node_off = (torch.arange(B + 1, dtype=torch.int32) * n).to(dev)
    row_ptr = (torch.arange(N + 1, dtype=torch.int32) * deg).to(dev)
    local = torch.randint(0, n, (N, deg), generator=g, dtype=torch.int32)
    col_idx = (local + (torch.arange(N, dtype=torch.int32) // n * n).unsqueeze(1)).reshape(-1).contiguous().to(dev)
    x = torch.randn(N, D, generator=g).to(dev).to(torch.bfloat16)
    layers = []
    for _ in range(L):
        ly = dict(w0=torch.randn(D, D, generator=g) / (D * 8) ** 0.5, w1=torch.randn(D, D, generator=g) / D ** 0.5)
        for k in ("s0", "s1", "s2"):
            ly[k] = torch.rand(D, generator=g) + 0.5
        for k in ("t0", "t1", "t2"):
            ly[k] = torch.rand(D, generator=g) * 0.9 - 0.3
        layers.append(ly)
    net = FoldedWideGIN(layers, dev)

    def fused():
        return net.forward(node_off, row_ptr, col_idx, x, big=False)      # (every subgraph has 128 nodes: no block-tiled launches)

    def layerwise():
        rows = x
        for i in range(L):
            rows, _ = net.forward(node_off, row_ptr, col_idx, rows, num_layers=1, first_layer=i, big=False)
        return rows

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.iters):
            fn()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / args.iters

    rows_f, pooled = fused()
    rows_l = layerwise()
    torch.cuda.synchronize()
    if not int(os.environ.get("GCC_GINW_DBG", "0")):          # (ablation builds of the kernel compute garbage on purpose)
        assert net.check_status() == 0
        assert torch.equal(rows_f, rows_l), "fused and layerwise launches disagree"
        assert bool(torch.isfinite(pooled).all())
    ms_f, ms_l = timed(fused), timed(layerwise)
    flops_layer = 2.0 * nnz * D + 2.0 * N * (D * D * 2)
    bytes_layer = 2.0 * N * D * 2 + 4.0 * nnz + 4.0 * (N + 1) + 4.0 * (B + 1) + 2 * D * D * 2 + 6 * D * 4   # rows in + out, CSR, weights
    bytes_fused = 2.0 * N * D * 2 + 4.0 * nnz + 4.0 * (N + 1) + 4.0 * (B + 1) + L * (2 * D * D * 2 + 6 * D * 4) + 4.0 * B * (L + 1) * D
    out = {
        "metric": "GIN layer stack, bf16, TFLOP/s (algorithmic)", "unit": "TFLOP/s", "n_gpus": 1,
        "config": {"workload": "BASELINE configs[4]: GIN hid=256 layers=%d deg=%d bf16 on %d subgraphs x %d nodes, eval-mode "
                               "BatchNorm folded, 1xMI355X" % (L, deg, B, n), "nodes": N, "edges": nnz},
        "dtype": "bf16 storage, f32 accumulate", "data": "synthetic",
        "fused": {"ms": ms_f, "tflops": flops_layer * L / ms_f / 1e9, "node_layers_per_sec": N * L / ms_f * 1e3,
                  "hbm_algorithmic_gbs": bytes_fused / ms_f / 1e6,
                  "roofline": {"bound": "mfma", "achieved": flops_layer * L / ms_f / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": flops_layer * L / ms_f / 1e9 / PEAK_BF16_TFLOPS}},
        "layerwise": {"ms": ms_l, "ms_per_layer": ms_l / L, "tflops": flops_layer * L / ms_l / 1e9,
                      "roofline": {"bound": "hbm", "achieved": bytes_layer * L / ms_l / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": bytes_layer * L / ms_l / 1e6 / PEAK_HBM_GBS,
                                   "algorithmic_bytes_per_launch": bytes_layer}},
        "algorithmic_flops_per_layer": flops_layer, "iters": args.iters,
        "late_w0": bool(int(os.environ.get("GCC_GINW_LATE_W0", "0"))),
    }
    if args.phases:
        from gcc_amd import _cabi
        ticks = torch.zeros(16, dtype=torch.int64, device=dev)
        _cabi.load().gcc_ginw_debug_ticks(ticks.data_ptr())
        fused()
        torch.cuda.synchronize()
        _cabi.load().gcc_ginw_debug_ticks(None)
        t = ticks.cpu().numpy()
        names = ["rows-in", "neighbour-counts", "fragments+pool0", "aggregation", "linear0", "linear1", "rows-out"]
        out["fused_phase_us_per_subgraph"] = {nm: float(t[i]) / 100.0 / max(int(t[15]), 1) for i, nm in enumerate(names)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
