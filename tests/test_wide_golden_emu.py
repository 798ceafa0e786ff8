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
