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
    """BASELINE configs[4] at full size (4096 subgraphs x 128 nodes, 32 in-neighbours each, 8 layers).  The whole batch is
    too large for the oracle in seconds, but subgraphs are independent: 32 of the 4096, drawn at random, are cut out of the
    device's result and compared with oracle/gin_wide.py run on exactly those subgraphs (rows and pooled output, the
    tolerances of the small-batch test).  Plus size-independent properties over the full batch: one-layer launches == the
    fused launch bit for bit, the pooled output is the per-subgraph sum of the rows, a permutation of the subgraphs
    permutes the results."""
    from gcc_amd.gin_wide import FoldedWideGIN

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, n, deg, L = 4096, 128, 32, 8
    N = B * n
    rng = np.random.default_rng(5)
    layers = random_layers(rng, L)
    net = FoldedWideGIN([{k: torch.from_numpy(v) for k, v in ly.items()} for ly in layers], dev)
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
    # 32 subgraphs of the full-size launch against the oracle (gin.py:42-58,213-232 at width 256, eval-mode BN folded)
    pick = np.sort(np.random.default_rng(17).choice(B, size=32, replace=False))
    sub_local = local.view(B, n, deg)[torch.from_numpy(pick)].numpy()                       # [32, n, deg] in-block ids
    o_node_off = (np.arange(33) * n).astype(np.int32)
    o_row_ptr = (np.arange(32 * n + 1) * deg).astype(np.int32)
    o_col = (sub_local + (np.arange(32) * n)[:, None, None]).reshape(-1).astype(np.int32)
    x_host = x.float().cpu().numpy().reshape(B, n, D)[pick].reshape(32 * n, D)
    want_rows, want_pooled = ow.gin_wide_forward(o_node_off, o_row_ptr, o_col, x_host, layers, bf16=True)
    got_rows = rows.float().cpu().numpy().reshape(B, n, D)[pick].reshape(32 * n, D)
    got_pooled = pooled.cpu().numpy()[pick]
    e_rows, e_pool = rel_err(got_rows, want_rows), rel_err(got_pooled, want_pooled)
    print("configs[4] full size, 32 of 4096 subgraphs vs oracle: rows %.2e pooled %.2e" % (e_rows, e_pool))
    assert e_rows < 1e-2 and e_pool < 5e-3
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


def test_subgraphs_of_129_to_1024_nodes_run_block_by_block():
    """ego-nets of the pre-training workload reach several hundred nodes: with the scratch that
    FoldedWideGIN passes by default, subgraphs over 128 nodes go through gin_wide_big_kernel -- (subgraph, 128-row block)
    work items, one launch per layer, the adjacency strip 128 columns at a time -- at the oracle's tolerance for small ones;
    the small subgraphs of the same batch are bit for bit what the fused launch gives without scratch."""
    from gcc_amd.gin_wide import FoldedWideGIN

    sizes = [129, 64, 1024, 300, 5, 128, 513, 256, 700, 0, 90]
    got, pooled, (e_rows, e_pool, e_truth) = _run(sizes, 12, 4, seed=11)
    assert e_rows < 5e-3 and e_pool < 2e-3 and e_truth < 3e-2 and np.isfinite(got).all()
    # per subgraph: no block left behind
    rng = np.random.default_rng(11)
    layers = random_layers(rng, 4)
    node_off, row_ptr, col_idx = random_batch(rng, sizes, 12)
    x = ow.bf16_round(rng.standard_normal((int(node_off[-1]), D)).astype(np.float32))
    want_rows, want_pooled = ow.gin_wide_forward(node_off, row_ptr, col_idx, x, layers, bf16=True)
    for b, n in enumerate(sizes):
        lo, hi = node_off[b], node_off[b + 1]
        if n:
            assert rel_err(got[lo:hi], want_rows[lo:hi]) < 5e-3, (b, n)
            assert rel_err(pooled[b], want_pooled[b]) < 2e-3, (b, n)
    # without scratch the big ones are refused (status bit 32) and the small ones are unchanged
    dev = torch.device("cuda:0")
    net = FoldedWideGIN([{k: torch.from_numpy(v) for k, v in ly.items()} for ly in layers], dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    rows0, pooled0 = net.forward(t(node_off), t(row_ptr), t(col_idx), t(x).to(torch.bfloat16), big=False)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="more than 128 nodes"):
        net.check_status()
    r0 = rows0.float().cpu().numpy()
    for b, n in enumerate(sizes):
        if 0 < n <= 128:
            lo, hi = node_off[b], node_off[b + 1]
            assert np.array_equal(r0[lo:hi], got[lo:hi]) and np.array_equal(pooled0[b].cpu().numpy(), pooled[b])
