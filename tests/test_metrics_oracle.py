"""The SSIM restatement against scipy's uniform_filter formulation (what skimage.metrics.structural_similarity computes)."""
import numpy as np

from oracle import metrics_oracle as MO


def test_ssim_sliding_window_equals_uniform_filter_formulation():
    rng = np.random.RandomState(0)
    for shape in ((32, 32), (64, 64), (9, 13)):
        a = rng.rand(*shape)
        b = np.clip(a + 0.1 * rng.randn(*shape), 0, 1)
        assert abs(MO.ssim_plane(a, b) - MO.ssim_plane_reference(a, b)) < 1e-12
    assert abs(MO.ssim_plane(a, a) - 1.0) < 1e-12


def test_sequence_metrics_known_values():
    t = np.zeros((3, 3, 16, 16))
    g = t + 0.2
    mse, psnr, ssim = MO.sequence_metrics(g, t)
    assert abs(mse - 0.04) < 1e-12
    assert abs(psnr - 10 * np.log10(1 / 0.01)) < 1e-9        # on [0, 1] images the offset is 0.1
    assert 0 < ssim < 1
