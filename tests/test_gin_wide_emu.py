"""Wide bf16 GIN layers (gcc_ginw_forward, BASELINE.json configs[4]) on the lock-step emulator vs oracle/gin_wide.py,
and the oracle's folded eval-mode algebra vs the torch modules of oracle/encoder.py."""
import numpy as np
import pytest
import torch

from oracle import gin_wide as ow
from oracle.encoder import _GIN
from tests.hipemu.emu_driver import emu_ginw_forward

D = 256


def random_layers(rng, L):
    layers = []
    for _ in range(L):
        ly = dict(w0=(rng.standard_normal((D, D)) / np.sqrt(D * 8)).astype(np.float32),
                  w1=(rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32))
        for k in ("s0", "s1", "s2"):
            ly[k] = rng.uniform(0.5, 1.5, D).astype(np.float32)
        for k in ("t0", "t1", "t2"):
            ly[k] = rng.uniform(-0.3, 0.6, D).astype(np.float32)
        layers.append(ly)
    return layers


def random_batch(rng, sizes, deg, symmetric=False):
    """in-neighbour lists: `deg` random in-block neighbours per node (duplicates allowed -> multi-edges)"""
    node_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    rows = []
    for b, n in enumerate(sizes):
        nb = [[] for _ in range(n)]
        for v in range(n):
            for u in rng.integers(0, n, size=min(deg, 4 * n)):
                nb[v].append(int(u))
                if symmetric:
                    nb[int(u)].append(v)
        rows += [np.asarray(sorted(r), dtype=np.int64) + node_off[b] for r in nb]
    row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    col_idx = (np.concatenate(rows) if len(rows) else np.zeros(0)).astype(np.int32)
    return node_off, row_ptr, col_idx


def bits_layers(layers):
    return [dict(ly, w0=ow.to_bf16_bits(ly["w0"]), w1=ow.to_bf16_bits(ly["w1"])) for ly in layers]


def rel_err(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def test_folded_algebra_matches_the_torch_modules():
    """oracle/gin_wide.py (float64, no rounding) == eval-mode ginlayers of oracle/encoder.py at hidden 256"""
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    gin = _GIN(3, D, D, D).eval()
    with torch.no_grad():
        for m in gin.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.5, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.4)
    node_off, row_ptr, col_idx = random_batch(rng, [9, 30, 1], 5, symmetric=True)
    N = int(node_off[-1])
    x = rng.standard_normal((N, D)).astype(np.float32)
    layers = []
    for i, layer in enumerate(gin.ginlayers):
        mlp = layer.apply_func.mlp
        bn = lambda m: (m.weight.detach().numpy(), m.bias.detach().numpy(), m.running_mean.numpy(), m.running_var.numpy())
        layers.append(ow.fold_layer(mlp.linears[0].weight.detach().numpy(), mlp.linears[0].bias.detach().numpy(),
                                    bn(mlp.batch_norms[0]), mlp.linears[1].weight.detach().numpy(),
                                    mlp.linears[1].bias.detach().numpy(), bn(layer.apply_func.bn), bn(gin.batch_norms[i])))
    got, pooled = ow.gin_wide_forward(node_off, row_ptr, col_idx, x, layers, bf16=False)
    # the layer loop of OracleGraphEncoder.forward (oracle/encoder.py; gin.py:217-221) on these modules
    h = torch.from_numpy(x).double()
    gin = gin.double()
    src = torch.repeat_interleave(torch.arange(N), torch.from_numpy(np.diff(row_ptr)).long())
    dst = torch.from_numpy(col_idx).long()
    F = torch.nn.functional
    with torch.no_grad():
        for i, layer in enumerate(gin.ginlayers):
            neigh = torch.zeros_like(h).index_add_(0, dst, h[src])
            z = (1 + layer.eps) * h + neigh
            mlp = layer.apply_func.mlp
            z = mlp.linears[1](F.relu(mlp.batch_norms[0](mlp.linears[0](z))))
            z = F.relu(layer.apply_func.bn(z))
            h = F.relu(gin.batch_norms[i](z))
    assert rel_err(got, h.numpy()) < 1e-5
    gid = np.repeat(np.arange(3), np.diff(node_off))
    want = np.zeros((3, D))
    np.add.at(want, gid, h.numpy())
    assert rel_err(pooled[:, -1], want) < 1e-5


@pytest.mark.parametrize("sizes,deg,L", [([128, 37, 1, 0, 64], 32, 2), ([5], 3, 1), ([90, 100, 17, 33, 113], 6, 3)])
def test_emulated_kernel_matches_the_oracle(sizes, deg, L):
    rng = np.random.default_rng(len(sizes) + L)
    layers = random_layers(rng, L)
    node_off, row_ptr, col_idx = random_batch(rng, sizes, deg)
    N = int(node_off[-1])
    x = ow.bf16_round(rng.standard_normal((N, D)).astype(np.float32))
    rows, pooled, status = emu_ginw_forward(node_off, row_ptr, col_idx, ow.to_bf16_bits(x), bits_layers(layers))
    assert status == 0
    # the fragment-major weight copies (gcc_ginw_pack_weights) hold the same numbers in request order: same results, bit for bit
    rows_p, pooled_p, status_p = emu_ginw_forward(node_off, row_ptr, col_idx, ow.to_bf16_bits(x), bits_layers(layers), pack=True)
    assert status_p == 0 and np.array_equal(rows, rows_p) and np.array_equal(pooled, pooled_p)
    want_rows, want_pooled = ow.gin_wide_forward(node_off, row_ptr, col_idx, x, layers, bf16=True)
    got = ow.from_bf16_bits(rows)
    # same rounding points, f32 (matrix core) vs f64 accumulation: a few results land on the other side of a bf16
    # rounding boundary (2^-8 relative) and perturb what follows; identical otherwise
    assert rel_err(got, want_rows) < 2e-3
    assert np.max(np.abs(got - want_rows)) <= 2.0 ** -6 * np.max(np.abs(want_rows))
    assert rel_err(pooled, want_pooled) < 1e-3
    for b, n in enumerate(sizes):
        if n == 0:
            assert not pooled[b].any()
    # bf16 storage vs unrounded float64 arithmetic with the same (bf16) weights
    truth, _ = ow.gin_wide_forward(node_off, row_ptr, col_idx, x,
                                   [dict(ly, w0=ow.bf16_round(ly["w0"]), w1=ow.bf16_round(ly["w1"])) for ly in layers], bf16=False)
    assert rel_err(got, truth) < 2e-2


@pytest.mark.parametrize("sizes,deg,L,pack", [([129, 40], 6, 2, True), ([300, 7, 128, 257], 5, 2, False)])
def test_subgraphs_over_128_nodes_run_block_by_block(sizes, deg, L, pack):
    """with scratch, subgraphs over GCC_GINW_MAX_NODES go through gin_wide_big_kernel: one launch per layer, one
    (subgraph, 128-row block) per workgroup, the adjacency strip taken 128 columns at a time with the products
    accumulating in registers -- the same rounding points as small subgraphs, so the same oracle at the same tolerance;
    small subgraphs of the same batch keep the fused launch and are bit for bit what they are without scratch."""
    rng = np.random.default_rng(sum(sizes) + L)
    layers = random_layers(rng, L)
    node_off, row_ptr, col_idx = random_batch(rng, sizes, deg)
    N = int(node_off[-1])
    x = ow.bf16_round(rng.standard_normal((N, D)).astype(np.float32))
    rows, pooled, status = emu_ginw_forward(node_off, row_ptr, col_idx, ow.to_bf16_bits(x), bits_layers(layers), pack=pack,
                                            scratch=True)
    assert status == 0
    want_rows, want_pooled = ow.gin_wide_forward(node_off, row_ptr, col_idx, x, layers, bf16=True)
    got = ow.from_bf16_bits(rows)
    for b, n in enumerate(sizes):
        lo, hi = node_off[b], node_off[b + 1]
        assert rel_err(got[lo:hi], want_rows[lo:hi]) < 2e-3, (b, n)
        assert np.max(np.abs(got[lo:hi] - want_rows[lo:hi])) <= 2.0 ** -6 * np.max(np.abs(want_rows)), (b, n)
        assert rel_err(pooled[b], want_pooled[b]) < 1e-3, (b, n)
    plain_rows, plain_pooled, plain_status = emu_ginw_forward(node_off, row_ptr, col_idx, ow.to_bf16_bits(x), bits_layers(layers),
                                                              pack=pack)
    assert plain_status == 32                                # without scratch the big ones are refused, as before
    for b, n in enumerate(sizes):
        if n <= 128:
            lo, hi = node_off[b], node_off[b + 1]
            assert np.array_equal(rows[lo:hi], plain_rows[lo:hi]) and np.array_equal(pooled[b], plain_pooled[b])


def test_input_pooling_and_refusals():
    rng = np.random.default_rng(7)
    layers = random_layers(rng, 1)
    node_off, row_ptr, col_idx = random_batch(rng, [20, 130], 4)
    N = int(node_off[-1])
    x = ow.bf16_round(rng.standard_normal((N, D)).astype(np.float32))
    rows, pooled, status = emu_ginw_forward(node_off, row_ptr, col_idx, ow.to_bf16_bits(x), bits_layers(layers))
    assert status == 32                                     # the 130-node subgraph is refused, loudly
    assert not rows[20:].any() and not pooled[1].any()
    np.testing.assert_allclose(pooled[0, 0], x[:20].astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    want_rows, _ = ow.gin_wide_forward(node_off[:2], row_ptr[:21], col_idx[:row_ptr[20]], x[:20], layers, bf16=True)
    assert rel_err(ow.from_bf16_bits(rows[:20]), want_rows) < 2e-3
    # a neighbour outside its subgraph is skipped and flagged
    node_off, row_ptr, col_idx = random_batch(rng, [6, 6], 2)
    col_idx = col_idx.copy()
    col_idx[0] = 9
    _, _, status = emu_ginw_forward(node_off, row_ptr, col_idx, ow.to_bf16_bits(x[:12]), bits_layers(layers))
    assert status == 64
    with pytest.raises(RuntimeError):                       # at least one layer
        emu_ginw_forward(node_off, row_ptr, col_idx, ow.to_bf16_bits(x[:12]), [])


def test_host_side_folding_equals_the_oracle_folding():
    """gcc_amd.gin_wide.fold_bn (what FoldedWideGIN.from_gin applies to a module with the reference's attribute names)
    == oracle/gin_wide.fold_layer; the product refuses to run without a GPU"""
    from gcc_amd.gin_wide import FoldedWideGIN, fold_bn

    torch.manual_seed(1)
    gin = _GIN(2, D, D, D).eval()
    with torch.no_grad():
        for m in gin.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.5, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.4)
    layer = gin.ginlayers[0]
    mlp = layer.apply_func.mlp
    bn = lambda m: (m.weight.detach().numpy(), m.bias.detach().numpy(), m.running_mean.numpy(), m.running_var.numpy())
    want = ow.fold_layer(mlp.linears[0].weight.detach().numpy(), mlp.linears[0].bias.detach().numpy(), bn(mlp.batch_norms[0]),
                         mlp.linears[1].weight.detach().numpy(), mlp.linears[1].bias.detach().numpy(), bn(layer.apply_func.bn),
                         bn(gin.batch_norms[0]))
    for (s, t), (ks, kt) in zip((fold_bn(mlp.batch_norms[0], mlp.linears[0].bias), fold_bn(layer.apply_func.bn, mlp.linears[1].bias),
                                 fold_bn(gin.batch_norms[0])), (("s0", "t0"), ("s1", "t1"), ("s2", "t2"))):
        np.testing.assert_allclose(s.numpy(), want[ks], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(t.numpy(), want[kt], rtol=1e-6, atol=1e-7)
    with pytest.raises(RuntimeError, match="GPU only"):
        FoldedWideGIN.from_gin(gin, "cpu")


def test_pack_weights_layout_is_the_one_the_header_documents():
    """include/gcc_amd.h, gcc_ginw_pack_weights: fragment (w, m, ks) is 1 KiB contiguous at ((w * 4 + m) * 8 + ks) * 512
    elements, lane (16 lg + lr) holding W[row][32 ks + 8 lg .. + 7]; which = 0: row = 64 w + 32 (m / 2) + 2 lr + m % 2,
    which = 1: row = 64 w + 16 m + lr."""
    from tests.hipemu.emu_driver import emu_lib

    lib = emu_lib()
    rng = np.random.default_rng(3)
    w = rng.integers(0, 1 << 16, size=(D, D), dtype=np.uint16)
    for which in (0, 1):
        got = np.zeros(D * D, dtype=np.uint16)
        assert lib.gcc_ginw_pack_weights(w.ctypes.data, got.ctypes.data, which, None) == 0
        want = np.zeros(D * D, dtype=np.uint16)
        for wb in range(4):
            for m in range(4):
                for ks in range(8):
                    base = ((wb * 4 + m) * 8 + ks) * 512
                    for lg in range(4):
                        for lr in range(16):
                            row = 64 * wb + (32 * (m // 2) + 2 * lr + m % 2 if which == 0 else 16 * m + lr)
                            lane = 16 * lg + lr
                            want[base + 8 * lane: base + 8 * lane + 8] = w[row, 32 * ks + 8 * lg: 32 * ks + 8 * lg + 8]
        np.testing.assert_array_equal(got, want)
    assert lib.gcc_ginw_pack_weights(None, got.ctypes.data, 0, None) != 0          # bad arguments are refused
    assert lib.gcc_ginw_pack_weights(w.ctypes.data, got.ctypes.data, 2, None) != 0
