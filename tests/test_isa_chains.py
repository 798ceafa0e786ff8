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
