"""CPU: the oracle (oracle/fcvsr_oracle.py) against golden vectors produced by the reference model itself."""
import pytest
import torch

from helpers import CASES, load_case, weights_for
from oracle import fcvsr_oracle as O

TOL = 2e-5


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_taps(name):
    x, gold, meta = load_case(name)
    p = weights_for(meta)
    taps = {}
    with torch.no_grad():
        y = O.forward(p, x, taps)
    assert y.shape == gold["out"].shape
    worst = {}
    for k, g in gold.items():
        assert k in taps, f"oracle has no tap {k}"
        t = taps[k]
        assert t.shape == g.shape, (k, t.shape, g.shape)
        scale = max(1.0, float(g.abs().max()))
        worst[k] = float((t - g).abs().max()) / scale
    bad = {k: v for k, v in worst.items() if v > TOL}
    assert not bad, f"taps beyond {TOL}: {bad}"
    assert float((y - gold["out"]).abs().max()) <= 1e-5


def test_oracle_etc_matches_reference():
    """GShiftNet_ETC (13 frames -> 7 SR frames + 7 bilinear bases) against the reference's own output (etc_12x16.npz)."""
    x, gold, meta = load_case("etc_12x16")
    with torch.no_grad():
        out_seq, x_up = O.forward_etc(weights_for(meta), x)
    assert out_seq.shape == gold["out_seq"].shape and x_up.shape == gold["x_up"].shape
    assert float((out_seq - gold["out_seq"]).abs().max()) <= 1e-5
    assert float((x_up - gold["x_up"]).abs().max()) <= 1e-6


def test_psnr_constants():
    """Known-answer PSNR values the reference's own tests hold (mmedit_train/tests/test_metrics/test_metrics.py:31-71)."""
    import numpy as np
    from fcvsr_amd.harness.metrics import psnr
    a = np.ones((32, 32), dtype=np.float64)
    assert abs(psnr(a, a * 2, crop_border=0) - 48.1308036) < 1e-6
    assert psnr(a, a, crop_border=0) == float("inf")
    assert psnr(np.zeros((32, 32)), np.full((32, 32), 255.0), crop_border=0) == 0
