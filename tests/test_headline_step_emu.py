"""Emulator tier of tests/test_headline_parity_gpu.py: one fused MoCo step and one fused E2E step on a freshly sampled
batch (C sampler oracle + SciPy positional embedding standing in for the device producer) through the SAME host code
and kernels (wave64 emulator build) against oracle/encoder.py -- checks the shared checker itself at a small size.
Reference: train.py:378-434."""
import numpy as np
import pytest
import torch

from gcc_amd.contrast import MemoryMoCo
from gcc_amd.graphgen import powerlaw_graph
from gcc_amd.train_step import E2ETrainStep, MoCoTrainStep
from oracle import posemb as P
from oracle import sampler as O
from tests.headline_step_check import check_e2e_step, check_moco_step
from tests.hipemu.emu_encoder import CpuBatch, emu_engine, reference_encoder
from tests.test_nce_emu import emu_nce

B, K = 24, 96


class OracleSampler:
    """batches from the C sampler oracle + the SciPy positional embedding (test stand-in for the device producer)."""
    batch_size = B

    def __init__(self):
        rp, ci = powerlaw_graph(3000, 30000, 1)
        c = O.COracle()
        seeds = c.draw_seeds(O.seed_cdf(rp), 5, 0, B)
        L = O.max_nodes_table(int(np.diff(rp).max()), 32, 0.8)[np.diff(rp)[seeds]]
        views = []
        for view in range(2):
            r = c.sample_batch(rp, ci, seeds, L, view, 5, 0, O.restart_threshold(0.8))
            pos = P.batched_positional_embedding(r["node_off"], r["row_ptr"], r["col_idx"], 32, seed=view)
            views.append(CpuBatch(dict(node_off=torch.from_numpy(r["node_off"].astype(np.int64)),
                                       row_ptr=torch.from_numpy(r["row_ptr"].astype(np.int64)),
                                       col_idx=torch.from_numpy(r["col_idx"].astype(np.int64)),
                                       pos_undirected=torch.from_numpy(pos))))
        self.views = tuple(views)

    def sample(self, first_id, prof=None):
        return self.views


@pytest.mark.parametrize("warm_steps", [0, 2])
def test_fused_moco_step_on_a_sampled_batch_vs_oracle(warm_steps):
    """warm_steps > 0: the checked step is not the trainer's first (Adam moments, EMA weights and queue rows of earlier
    steps are part of the state the oracle starts from)."""
    torch.manual_seed(0)
    model, ema = reference_encoder(), reference_encoder()
    ema.load_state_dict(model.state_dict())
    model._engine = ema._engine = emu_engine()
    contrast = MemoryMoCo(64, None, K, 0.07, use_softmax=True)
    contrast._engine = emu_nce()
    tr = MoCoTrainStep(model, ema, contrast, OracleSampler(), posemb=lambda gr: gr, prefetch=False)
    tr.dropout_seed = 5
    for i in range(warm_steps):
        tr.step(i, 0.005)
    masks = (torch.rand(5, B, 64) >= 0.5).float().contiguous()
    rep = check_moco_step(tr, model, ema, contrast, 0.004, masks, step_id=warm_steps)
    assert rep["nodes_q"] > B and rep["loss_rel_err"] < 1e-3


def test_fused_e2e_step_on_a_sampled_batch_vs_oracle():
    torch.manual_seed(1)
    model = reference_encoder()
    model._engine = emu_engine()
    tr = E2ETrainStep(model, OracleSampler(), posemb=lambda gr: gr, prefetch=False, engine=emu_nce())
    mq = (torch.rand(5, B, 64) >= 0.5).float().contiguous()
    mk = (torch.rand(5, B, 64) >= 0.5).float().contiguous()
    rep = check_e2e_step(tr, model, 0.005, mq, mk)
    assert rep["nodes_q"] > B
