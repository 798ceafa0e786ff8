"""The dependent-load structure of the training stream's kernels, read off the gfx950 ISA (no GPU needed).

A short kernel's duration is the number of memory round trips the compiler left dependent on each other
(DESIGN.md 4b); the source does not show them, `tools/isa_chains.py` does.  These checks pin the property the second pass
of round 3 established -- a tile kernel's prologue (node count, statistics replicas or totals, BatchNorm weights, staged
Linear weights) is ONE batch of requests before its first barrier -- so that a refactoring which re-serialises it fails
here, on the CPU tier, instead of showing up as microseconds in a trace."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_chains  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(isa_chains.HIPCC) or shutil.which("make") is None,
                                reason="hipcc not installed")


def _chains(src):
    text = isa_chains.isa_of(isa_chains.ROOT / "gcc_amd" / "csrc" / src)
    out = {}
    for name, body in isa_chains.kernels(text):
        m = re.search(r"\d+([a-z0-9_]+_kernel(?:IL[a-z0-9_]+E)?)", name)
        out[m.group(1) if m else name] = isa_chains.chain(body)
    return out


def _waits_before_first_barrier(chain):
    head = chain.split("B", 1)[0]
    return sum(int(n or 1) for n in re.findall(r"W(\d*)", head))


def test_chain_notation_on_a_toy_listing():
    body = ["global_load_dword v1, v2, s[0:1]", "s_waitcnt vmcnt(0)", "global_load_dword v1, v2, s[0:1]",
            "global_load_dword v3, v2, s[0:1]", "s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(0)", "s_barrier"]
    assert isa_chains.chain(body) == "LWL2W2B"


def test_forward_tile_kernels_request_their_prologue_as_one_batch():
    ch = _chains("encoder.hip")
    for k in ("gin_mid_kernel", "gin_stat_kernel", "gin_pool_kernel"):
        # replicas (16) + weights, biases, running statistics + node count AND (round 6) the first tile's rows in flight together:
        # no load is requested after the first wait (the waits of one batch may come in steps: vmcnt(n) counting down)
        head = ch[k].split("B", 1)[0]
        assert "W" in head and "L" not in head[head.index("W"):], (k, ch[k])
        assert re.search(r"L([89]|[1-9][0-9])", head), (k, ch[k])              # the replica loads (kRep / 2 = 8 per thread) are one run
        assert sum(int(n or 1) for n in re.findall(r"L(\d*)", head)) >= 8 + 4 + 4, (k, ch[k])   # replicas + BatchNorm numbers + the tile's rows
    # no load-wait-load-wait ladders anywhere in the forward kernels' listings
    for k, v in ch.items():
        assert "LWLWLWLW" not in v, (k, v)


def test_backward_tile_kernels_have_a_one_batch_fast_path():
    ch = _chains("encoder_bwd.hip")
    # the fast arm (statistics' totals present) of the kernels with two coefficient tables: one run of requests -- two totals +
    # weight + bias per table, the backward sums' 16 replicas where the kernel has them, the node count, and (round 6) the
    # first tile's rows: 8 + 4 graph ids + row pointer + 8 rows in gin_bwd_c, 8 + 8 replicas + 8 rows in gin_bwd_b, ...
    assert re.search(r"L2[0-9]WB", ch["gin_bwd_c_kernel"]), ch["gin_bwd_c_kernel"]
    assert re.search(r"L2[0-9]WB", ch["gin_bwd_b_kernel"]), ch["gin_bwd_b_kernel"]
    assert re.search(r"L2[0-9]WB", ch["gin_bwd_lin_kernelILb1E"]) and re.search(r"L2[0-9]WB", ch["gin_bwd_lin_kernelILb0E"])
    for k, v in ch.items():
        assert "LWLWLWLW" not in v, (k, v)


def test_eval_kernels_keep_their_lds_pointers_and_their_weight_prefetch():
    """Round 6, csrc/encoder_eval.hip -- three properties of the LDS-resident eval kernel that the source does not show and the
    emulator cannot see:
    * no flat_load / flat_store: a lambda whose closure stays in memory (select chains over captured values become an indexed
      load from the closure object) turns every LDS pointer it captured into a generic one -- 180 flat instructions once;
    * the next layer's weights come by global_load_dwordx4 (the 16-byte and the element-wise request forms behind ONE run-time
      branch were merged into sixteen global_load_dword);
    * inside the layer loop no wait for ALL outstanding loads sits between the weight requests and the barrier behind them (a
      descriptor pointer indexed by a per-wave vector index came through global_load + s_waitcnt vmcnt(0): every layer waited
      for its successor's weights)."""
    text = isa_chains.isa_of(isa_chains.ROOT / "gcc_amd" / "csrc" / "encoder_eval.hip")
    bodies = {name: body for name, body in isa_chains.kernels(text)}
    lds = [b for n, b in bodies.items() if "gin_eval_lds_kernelILb1E" in n]
    assert len(lds) == 1
    body = lds[0]
    ops = [s.split()[0] for s in body if s and not s.startswith((";", "."))]
    assert not [o for o in ops if o.startswith(("flat_load", "flat_store"))]
    assert sum(o == "global_load_dwordx4" for o in ops) >= 4                     # 2 matrices x 2 quads per thread, in the loop
    # the layer loop: the stretch of the listing that holds the matrix instructions; find the weight requests inside it
    first_mfma = next(i for i, s in enumerate(body) if s.startswith("v_mfma"))
    x4 = [i for i, s in enumerate(body) if s.startswith("global_load_dwordx4")]
    in_loop = [i for i in x4 if i < first_mfma][-4:]                             # the last requests ahead of the products: the loop's
    assert len(in_loop) == 4 and in_loop[-1] - in_loop[0] < 40, (x4, first_mfma)
    barrier = next(i for i in range(in_loop[-1], len(body)) if body[i].startswith("s_barrier"))
    between = body[in_loop[-1]:barrier]
    assert not [s for s in between if re.match(r"s_waitcnt\s+vmcnt\(0\)", s)], between
    # and the spills stay out of the matrix-instruction stretch
    last_mfma = max(i for i, s in enumerate(body) if s.startswith("v_mfma"))
    assert not [s for s in body[first_mfma:last_mfma] if s.startswith("scratch_")]
