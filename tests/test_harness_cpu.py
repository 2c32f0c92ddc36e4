"""CPU: host logic of the evaluation harness - SSIM known answers of the reference's own metric tests, the streamed
multi-sequence scheduler (batching, padded last batch, double-buffer order, rank sharding) around a stand-in module."""
import numpy as np
import pytest
import torch


def test_ssim_known_answers_of_the_reference():
    """mmedit_train/tests/test_metrics/test_metrics.py:74-106: 0.9130623 for ones vs twos in every layout / crop, 0.9987801 on Y."""
    from fcvsr_amd.harness.metrics import ssim
    hw1, hw2 = np.ones((32, 32)), np.ones((32, 32)) * 2
    hwc1, hwc2 = np.ones((32, 32, 3)), np.ones((32, 32, 3)) * 2
    chw1, chw2 = np.ones((3, 32, 32)), np.ones((3, 32, 32)) * 2
    with pytest.raises(ValueError):
        ssim(hw1, hw2, crop_border=0, input_order="HH")
    with pytest.raises(ValueError):
        ssim(hw1, hw2, crop_border=0, input_order="ABC")
    np.testing.assert_almost_equal(ssim(hw1, hw2, crop_border=0), 0.9130623)
    np.testing.assert_almost_equal(ssim(hwc1, hwc2, crop_border=0, input_order="HWC"), 0.9130623)
    np.testing.assert_almost_equal(ssim(chw1, chw2, crop_border=0, input_order="CHW"), 0.9130623)
    np.testing.assert_almost_equal(ssim(hw1, hw2, crop_border=2), 0.9130623)
    np.testing.assert_almost_equal(ssim(hwc1, hwc2, crop_border=3, input_order="HWC"), 0.9130623)
    np.testing.assert_almost_equal(ssim(chw1, chw2, crop_border=4, input_order="CHW"), 0.9130623)
    np.testing.assert_almost_equal(ssim(hwc1, hwc2, crop_border=0, convert_to=None), 0.9130623)
    np.testing.assert_almost_equal(ssim(hwc1, hwc2, crop_border=0, convert_to="Y"), 0.9987801)


def test_ssim_separable_filter_equals_the_full_window():
    """The separable 'valid' filter is the 11x11 Gaussian window of the reference (psnr_ssim.py:333-343) on random images."""
    from scipy.ndimage import correlate
    from fcvsr_amd.harness.metrics import _filter_valid, _gaussian_window, ssim
    rs = np.random.RandomState(0)
    img = rs.rand(40, 52) * 255
    g = _gaussian_window()
    full = correlate(img, np.outer(g, g), mode="mirror")[5:-5, 5:-5]       # cv2.filter2D default border, then the 5-pixel crop
    assert np.abs(_filter_valid(img, g) - full).max() < 1e-9
    a = rs.randint(0, 256, (48, 64)).astype(np.uint8)
    assert ssim(a, a, crop_border=4) == pytest.approx(1.0)
    b = np.clip(a.astype(np.int32) + rs.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
    assert 0.0 < ssim(a, b, crop_border=4) < 1.0


class _Bilinear4(torch.nn.Module):
    """Stand-in with the drop-in call contract (B,7,C,H,W) -> (B,C,4H,4W): 4x bilinear of the centre frame plus the window mean."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))
        self.calls = []

    def forward(self, x):
        self.calls.append(tuple(x.shape))
        c = x[:, x.shape[1] // 2] * 0.5 + x.mean(1) * 0.5
        return torch.nn.functional.interpolate(c, scale_factor=4, mode="bilinear", align_corners=False) * self.w


@pytest.mark.parametrize("world", [1, 2, 3])
def test_streamed_scheduler_equals_sequence_by_sequence(world):
    from fcvsr_amd.harness.infer import StreamedSuperResolver, super_resolve_sequence
    rs = np.random.RandomState(1)
    seqs = [torch.from_numpy(rs.rand(n, 1, 10, 12).astype(np.float32)) for n in (9, 4, 6)]     # H = 10: padded to 12 inside
    model = _Bilinear4()
    ref = [super_resolve_sequence(model, s, batch=5) for s in seqs]
    got = {s: np.zeros_like(r) for s, r in enumerate(ref)}
    seen = {s: np.zeros(len(r), dtype=np.int32) for s, r in enumerate(ref)}
    for rank in range(world):
        model.calls.clear()
        res = StreamedSuperResolver(model, batch=4).run(seqs, rank=rank, world=world)
        assert all(c == (4, 7, 1, 12, 12) for c in model.calls)            # every call has the full batch shape
        for s, (first, arr) in res.items():
            got[s][first:first + len(arr)] = arr
            seen[s][first:first + len(arr)] += 1
    for s in range(len(seqs)):
        assert (seen[s] == 1).all()                                         # every frame exactly once over the ranks
        assert np.array_equal(got[s], ref[s])


def test_streamed_scheduler_rejects_mixed_frame_sizes():
    from fcvsr_amd.harness.infer import StreamedSuperResolver
    seqs = [torch.zeros(3, 1, 8, 8), torch.zeros(3, 1, 8, 12)]
    with pytest.raises(ValueError):
        StreamedSuperResolver(_Bilinear4(), batch=2).run(seqs)
