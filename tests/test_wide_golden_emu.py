"""Emulator tier of the width-128 / 256 reference-executed fixtures (tests/wide_golden_check.py)."""
import pytest

from tests.hipemu.emu_encoder import CpuBatch
from tests.test_wide_encoder_emu import emu_wide_engine, emu_wide_nce
from tests.wide_golden_check import gold, run_moco_step


def test_fixture_shapes_are_the_reference_models():
    from tests.test_wide_encoder_emu import wide_encoder

    for hidden, c in gold()["cases"].items():
        ours = dict(wide_encoder(hidden, hidden).named_parameters())
        assert set(c["grads64"]) <= set(ours) and all(ours[n].shape == g.shape for n, g in c["grads64"].items())
        assert c["out"].shape[1] == c["K"] + 1 and c["feat_q"].shape[1] == hidden
        assert c["ref_fp32_vs_f64"] < 1e-3


@pytest.mark.parametrize("hidden", [128, 256])
def test_moco_step_matches_the_reference_run(hidden, monkeypatch):
    worst = run_moco_step(hidden, "cpu", CpuBatch, monkeypatch, gin_engine=emu_wide_engine, nce_engine=emu_wide_nce)
    print(f"hidden {hidden}: worst gradient error vs the reference's float64 run {worst:.2e} of the tensor's largest entry")


class _FixtureSampler:
    def __init__(self, views, batch_size):
        self.views, self.batch_size = views, batch_size

    def sample(self, first_id, prof=None):
        return self.views


@pytest.mark.parametrize("hidden", [128, 256])
def test_fused_wide_step_reproduces_the_reference_post_step_state(hidden):
    """MoCoTrainStep at --hidden-size 128 / 256 (flat buffers, clip + Adam + EMA + meters as two launches, the any-width encoder
    and head underneath) against what the reference's train.py step left behind: loss, grad norm, weights, EMA, queue."""
    import torch

    from gcc_amd.contrast import MemoryMoCo
    from gcc_amd.train_step import MoCoTrainStep
    from tests.golden import wide_init
    from tests.test_nce_emu import emu_nce
    from tests.test_wide_encoder_emu import wide_encoder

    G = gold()
    c = G["cases"][hidden]
    model, ema = wide_init.fill_(wide_encoder(hidden, hidden), 0), wide_init.fill_(wide_encoder(hidden, hidden), 1)
    model._wide_engine = ema._wide_engine = emu_wide_engine()
    contrast = MemoryMoCo(hidden, None, c["K"], c["T"], use_softmax=True)
    with torch.no_grad():
        contrast.memory.copy_(wide_init.tensor_for("contrast.memory", contrast.memory) * c["memory0_scale"])
    contrast._engine = emu_wide_nce()
    views = (CpuBatch(G["views"][0]), CpuBatch(G["views"][1]))
    step = MoCoTrainStep(model, ema, contrast, _FixtureSampler(views, views[0].batch_size), posemb=lambda gr: gr, prefetch=False,
                         flat_engine=emu_nce())
    assert step.wide and not step.use_graph
    masks = c["masks"].contiguous()
    step.mask_fn = lambda: masks
    out = step.step(0, c["lr"])
    torch.testing.assert_close(out["loss"].reshape(()), c["loss"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out["prob"].reshape(()), c["prob"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(torch.as_tensor(out["grad_norm"]).reshape(()).float(), c["grad_norm"].float(), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(contrast.memory, c["after"]["memory"], rtol=1e-3, atol=2e-5)
    assert contrast.index == c["after"]["index"]
    for sd_name, mod in (("model", model), ("model_ema", ema)):
        sd = mod.state_dict()
        for key, ref in c["after"][sd_name].items():
            if ref.dtype.is_floating_point:
                torch.testing.assert_close(sd[key], ref, rtol=1e-4, atol=2e-6, msg=lambda m, key=key, n=sd_name: f"{n}.{key}: {m}")
