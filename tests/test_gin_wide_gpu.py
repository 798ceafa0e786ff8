"""gcc_ginw_forward on the MI355X (through gcc_amd.gin_wide -> C ABI) vs oracle/gin_wide.py."""
import numpy as np
import pytest
import torch

from oracle import gin_wide as ow
from tests.test_gin_wide_emu import D, random_batch, random_layers, rel_err

pytestmark = pytest.mark.gpu


def _run(sizes, deg, L, seed, split=None):
    from gcc_amd.gin_wide import FoldedWideGIN

    rng = np.random.default_rng(seed)
    layers = random_layers(rng, L)
    node_off, row_ptr, col_idx = random_batch(rng, sizes, deg)
    N = int(node_off[-1])
    x = ow.bf16_round(rng.standard_normal((N, D)).astype(np.float32))
    dev = torch.device("cuda:0")
    net = FoldedWideGIN([{k: torch.from_numpy(v) for k, v in ly.items()} for ly in layers], dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    xd = t(x).to(torch.bfloat16)
    if split is None:
        rows, pooled = net.forward(t(node_off), t(row_ptr), t(col_idx), xd)
    else:                                                # the same stack as separate launches (rows via HBM in between)
        rows, pooled = xd, None
        for first in range(0, L, split):
            rows, p = net.forward(t(node_off), t(row_ptr), t(col_idx), rows, num_layers=min(split, L - first), first_layer=first)
            pooled = p if pooled is None else torch.cat((pooled, p[:, 1:]), dim=1)
    torch.cuda.synchronize()
    assert net.check_status() == 0
    want_rows, want_pooled = ow.gin_wide_forward(node_off, row_ptr, col_idx, x, layers, bf16=True)
    truth, _ = ow.gin_wide_forward(node_off, row_ptr, col_idx, x,
                                   [dict(ly, w0=ow.bf16_round(ly["w0"]), w1=ow.bf16_round(ly["w1"])) for ly in layers], bf16=False)
    got = rows.float().cpu().numpy()
    errs = (rel_err(got, want_rows), rel_err(pooled.cpu().numpy(), want_pooled), rel_err(got, truth))
    print("wide GIN L=%d: rows vs bf16 oracle %.2e, pooled %.2e, rows vs float64 truth %.2e" % ((L,) + errs))
    return got, pooled.cpu().numpy(), errs


def test_eight_layers_match_the_oracle():
    # 300 subgraphs (> one per CU: the workgroups walk the batch), ragged sizes, multi-edges, an empty graph
    rng = np.random.default_rng(1)
    sizes = [128, 1, 0, 37] + [int(s) for s in rng.integers(2, 129, size=296)]
    got, pooled, (e_rows, e_pool, e_truth) = _run(sizes, 32, 8, seed=2)
    # tolerance of the bf16 mode: identical rounding points, f32 vs f64 accumulation -> isolated bf16 rounding flips
    # (2^-8 relative each) that the following layers carry along
    assert e_rows < 1e-2 and e_pool < 5e-3
    assert e_truth < 5e-2                                # eight layers of bf16 storage vs unrounded arithmetic
    assert np.isfinite(got).all()


def test_layerwise_launches_equal_the_fused_launch():
    sizes = [128, 64, 9, 100] * 8
    a, pa, _ = _run(sizes, 32, 4, seed=3)
    b, pb, _ = _run(sizes, 32, 4, seed=3, split=1)
    np.testing.assert_array_equal(a, b)                  # rows are bf16 in LDS exactly as in HBM
    np.testing.assert_array_equal(pa, pb)


def test_config5_full_size_properties():
    """BASELINE configs[4] at full size (4096 subgraphs x 128 nodes, 32 in-neighbours each, 8 layers): too large for the
    oracle in seconds, so size-independent properties: one-layer launches == the fused launch bit for bit, the pooled
    output is the per-subgraph sum of the rows, a permutation of the subgraphs permutes the results."""
    from gcc_amd.gin_wide import FoldedWideGIN

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, n, deg, L = 4096, 128, 32, 8
    N = B * n
    rng = np.random.default_rng(5)
    net = FoldedWideGIN([{k: torch.from_numpy(v) for k, v in ly.items()} for ly in random_layers(rng, L)], dev)
    node_off = (torch.arange(B + 1, dtype=torch.int32) * n).to(dev)
    row_ptr = (torch.arange(N + 1, dtype=torch.int32) * deg).to(dev)
    local = torch.randint(0, n, (N, deg), generator=g, dtype=torch.int32)
    base = (torch.arange(N, dtype=torch.int32) // n * n).unsqueeze(1)
    col_idx = (local + base).reshape(-1).contiguous().to(dev)
    x = torch.randn(N, D, generator=g).to(dev).to(torch.bfloat16)
    rows, pooled = net.forward(node_off, row_ptr, col_idx, x)
    step = x
    for i in range(L):
        step, _ = net.forward(node_off, row_ptr, col_idx, step, num_layers=1, first_layer=i)
    torch.cuda.synchronize()
    assert net.check_status() == 0
    assert torch.equal(rows, step)
    assert bool(torch.isfinite(pooled).all()) and float(rows.float().abs().max()) > 0
    sums = rows.float().view(B, n, D).sum(1)
    assert float((pooled[:, -1] - sums).abs().max()) <= 1e-3 * float(sums.abs().max())
    # subgraphs are independent: reversing their order reverses the results and changes nothing else
    perm = torch.arange(B - 1, -1, -1, device=dev)
    xp = x.view(B, n, D)[perm].reshape(N, D).contiguous()
    localp = local.view(B, n, deg)[perm.cpu()].reshape(N, deg)
    colp = (localp + base).reshape(-1).contiguous().to(dev)
    rows_p, pooled_p = net.forward(node_off, row_ptr, colp, xp)
    torch.cuda.synchronize()
    assert torch.equal(rows_p.view(B, n, D)[perm].reshape(N, D), rows)
    assert torch.equal(pooled_p[perm], pooled)
