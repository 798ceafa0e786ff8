"""Per-worker-shard sampling against values recorded by EXECUTING the reference's own
``LoadBalanceGraphDataset.__init__`` / ``__iter__`` (tests/golden/make_shard_golden.py -> tests/golden/shard_reference.json): the
greedy assignment of graphs to workers and, per worker, the seed distribution in_degree ** 0.75 / sum that it hands to
``np.random.choice``.  Checked here: the layout the shard tests use (tests/shard_check.py, which the emulator and GPU tiers compare
the kernel with) and the oracle's per-shard seed cdf (``oracle/sampler.py: seed_cdf``) reproduce both."""
import json
import os

import numpy as np

from oracle import sampler as O
from tests.shard_check import corpus, reference_layout

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shard_reference.json")


def test_jobs_and_per_shard_seed_distribution_are_the_reference_ones():
    gold = json.load(open(GOLD))
    graphs = corpus()
    assert len(gold["configs"]) == 4
    for cfg in gold["configs"]:
        jobs, rp, ci, shard_off = reference_layout(graphs, cfg["num_workers"], cfg["num_copies"])
        assert jobs == cfg["jobs"], cfg["num_workers"]
        nshard = len(shard_off) - 1
        cdf = O.seed_cdf(rp, shard_off if nshard > 1 else None)
        for w, sh in enumerate(cfg["shards"]):
            s = w % nshard                                     # jobs * num_copies: worker w holds shard w % bins
            assert sh["jobs"] == jobs[w]
            lo, hi = int(shard_off[s]), int(shard_off[s + 1])
            assert hi - lo == sh["length"]
            c = np.asarray(cdf[lo:hi], dtype=np.float64)       # cumulative inside the shard, ending at 1
            p = np.diff(np.concatenate([[0.0], c]))
            assert abs(c[-1] - 1.0) < 1e-12
            np.testing.assert_allclose(p[:64], sh["p_head"], rtol=1e-10, atol=1e-15)
            assert abs((p ** 2).sum() - sh["p_sum_sq"]) < 1e-12 * max(sh["p_sum_sq"], 1e-30) + 1e-18
            # (several nodes share the largest degree: differencing the cdf decides the tie by rounding noise, so the reference's
            #  argmax is looked up rather than recomputed)
            assert abs(p.max() - sh["p_max"]) < 1e-12 and abs(p[sh["p_argmax"]] - sh["p_max"]) < 1e-12
