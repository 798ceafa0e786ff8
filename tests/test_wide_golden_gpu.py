"""Device tier of the width-128 / 256 reference-executed fixtures (tests/wide_golden_check.py): one MoCo step of the any-width
path on cuda:0 against what the reference's GraphEncoder / MemoryMoCo / loss / Adam produced at --hidden-size 128 and 256."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden", [128, 256])
def test_moco_step_on_the_device_matches_the_reference_run(hidden, monkeypatch):
    from tests.test_encoder_gpu import gpu_batch
    from tests.wide_golden_check import run_moco_step

    worst = run_moco_step(hidden, "cuda:0", gpu_batch, monkeypatch)
    print(f"hidden {hidden}: worst gradient error vs the reference's float64 run {worst:.2e} of the tensor's largest entry")
