"""CPU oracle of the wide (hidden 256) bf16 GIN layer stack of include/gcc_amd.h `gcc_ginw_forward` -- TEST
INFRASTRUCTURE ONLY (numpy; nothing under gcc_amd/ imports it).

Restates, per layer, UnsupervisedGIN.forward gcc/models/gin.py:217-221 with the modules in eval mode:
    DGL GINConv(sum, eps=0)      call site gin.py:179-185   agg = h + sum over the in-neighbours
    MLP.forward                  gin.py:113-116             linears.1(relu(batch_norms.0(linears.0(agg))))
    ApplyNodeFunc.forward        gin.py:55-57               relu(bn(mlp(.)))
    outer BatchNorm + ReLU       gin.py:219-220
    SumPooling of hidden_rep     gin.py:205,228
`fold_layer` turns the Linear biases and BatchNorm running statistics into per-channel scale/shift pairs.

Parity status: the folded algebra is pinned by tests/test_gin_wide_emu.py against oracle/encoder.py's torch
modules (themselves pinned by golden vectors produced by executing the reference's gin.py).  The bf16 storage
points (agg, z1, h rounded to nearest even) are this framework's own: the reference has no bf16 path, so for
them parity is UNPINNED and the tests bound the distance to the unrounded float64 result instead.
"""
from __future__ import annotations

import numpy as np


def to_bf16_bits(x):
    """float32 -> bf16 bit pattern (uint16), round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return (u >> 16).astype(np.uint16)


def from_bf16_bits(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def bf16_round(x):
    return from_bf16_bits(to_bf16_bits(x))


def fold_layer(lin0_w, lin0_b, bn0, lin1_w, lin1_b, bn_a, bn_c, eps=1e-5):
    """bn* = (weight, bias, running_mean, running_var) of mlp.batch_norms.0, apply_func.bn, gin.batch_norms[i].
    Returns float64/float32 arrays: w0, w1 (unrounded) and s0, t0, s1, t1, s2, t2 (float32)."""
    def st(bn, bias):
        g, b, mu, var = (np.asarray(t, dtype=np.float64) for t in bn)
        s = g / np.sqrt(var + eps)
        return s, (np.asarray(bias, dtype=np.float64) - mu) * s + b
    s0, t0 = st(bn0, lin0_b)
    s1, t1 = st(bn_a, lin1_b)
    s2, t2 = st(bn_c, np.zeros_like(np.asarray(lin1_b, dtype=np.float64)))
    f = np.float32
    return dict(w0=np.asarray(lin0_w, dtype=f), w1=np.asarray(lin1_w, dtype=f), s0=s0.astype(f), t0=t0.astype(f),
                s1=s1.astype(f), t1=t1.astype(f), s2=s2.astype(f), t2=t2.astype(f))


def gin_wide_forward(node_off, row_ptr, col_idx, x, layers, bf16=True):
    """x: float32 [N, 256] (already bf16-representable when bf16=True).  Row v of the CSR lists v's in-neighbours
    (global ids).  Returns (h [N, 256] float32, pooled [B, L + 1, 256] float32).  With bf16=False nothing is
    rounded and the arithmetic is float64 (the 'truth' the bf16 path is bounded against)."""
    node_off = np.asarray(node_off, dtype=np.int64)
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    col_idx = np.asarray(col_idx, dtype=np.int64)
    N, B, L = int(node_off[-1]), len(node_off) - 1, len(layers)
    rnd = bf16_round if bf16 else (lambda v: v)
    h = np.asarray(x, dtype=np.float64)[:N]
    D = h.shape[1]
    dst = np.repeat(np.arange(N), np.diff(row_ptr)[:N])
    gid = np.repeat(np.arange(B), np.diff(node_off))
    pooled = np.zeros((B, L + 1, D), dtype=np.float64)
    np.add.at(pooled[:, 0], gid, h)
    for i, ly in enumerate(layers):
        w0 = (rnd(ly["w0"]) if bf16 else ly["w0"]).astype(np.float64)
        w1 = (rnd(ly["w1"]) if bf16 else ly["w1"]).astype(np.float64)
        agg = h.copy()
        np.add.at(agg, dst, h[col_idx[:len(dst)]])
        agg = rnd(agg.astype(np.float32)).astype(np.float64) if bf16 else agg
        z1 = np.maximum((agg @ w0.T) * ly["s0"].astype(np.float64) + ly["t0"], 0.0)
        z1 = rnd(z1.astype(np.float32)).astype(np.float64) if bf16 else z1
        y = np.maximum((z1 @ w1.T) * ly["s1"].astype(np.float64) + ly["t1"], 0.0)
        h = np.maximum(y * ly["s2"].astype(np.float64) + ly["t2"], 0.0)
        h = rnd(h.astype(np.float32)).astype(np.float64) if bf16 else h
        np.add.at(pooled[:, i + 1], gid, h)
    return h.astype(np.float32), pooled.astype(np.float32)
