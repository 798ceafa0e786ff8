"""Generates tests/golden/sampler_golden.json with the PURE-PYTHON restatement
(oracle/sampler.py py_*), i.e. independently of both the C oracle and the HIP
kernels.  Run from the repo root:  python tests/golden/make_sampler_golden.py

The reference holds no golden vectors for the sampler; these are
authored here from the spec in oracle/sampler_oracle.c.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gcc_amd.graphgen import powerlaw_graph, tiny_graphs  # noqa: E402
from oracle import sampler as O  # noqa: E402


def case(name, rp, ci, seed, L, run_seed, g, restart_prob=0.8):
    thr = O.restart_threshold(restart_prob)
    trace = O.py_rwr_trace(rp, ci, seed, L, run_seed, g, thr)
    nodes, sub_rp, sub_col = O.py_subgraph(rp, ci, seed, trace)
    return dict(name=name, seed=seed, L=L, run_seed=run_seed, g=g, restart_prob=restart_prob,
                trace=trace, nodes=nodes, sub_row_ptr=sub_rp, sub_col=sub_col)


def main():
    out = {"philox_kat": [
        dict(ctr=[0, 0, 0, 0], key=[0, 0], out=[0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        dict(ctr=[0xffffffff] * 4, key=[0xffffffff] * 2, out=[0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        dict(ctr=[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], key=[0xa4093822, 0x299f31d0],
             out=[0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ], "graphs": {}, "cases": []}
    for name, (rp, ci) in tiny_graphs().items():
        out["graphs"][name] = dict(row_ptr=rp.tolist(), col_idx=ci.tolist())
        for seed in (0, len(rp) - 2):
            for g in (0, 1, 7):
                out["cases"].append(dict(graph=name, **case(name, rp, ci, seed, 12, 5, g)))
    rp, ci = powerlaw_graph(400, 3000, 9)
    out["graphs"]["pl400"] = dict(row_ptr=rp.tolist(), col_idx=ci.tolist())
    cdf = O.seed_cdf(rp)
    out["pl400_seeds"] = dict(run_seed=21, first=1000, seeds=[O.py_draw_seed(cdf, 21, 1000 + i) for i in range(32)])
    for i, seed in enumerate(out["pl400_seeds"]["seeds"][:6]):
        out["cases"].append(dict(graph="pl400", **case("pl400", rp, ci, seed, 64, 21, (1000 + i) * 2 + (i & 1))))
    with open(os.path.join(os.path.dirname(__file__), "sampler_golden.json"), "w") as f:
        json.dump(out, f)
    print("cases:", len(out["cases"]))


if __name__ == "__main__":
    main()
