"""ORACLE (test infrastructure only — never imported by the product path).

numpy restatement of the evaluation metrics the reference's Evaluator calls (gcp/evaluation/compute_metrics.py:123-130):
`mse`, `psnr`, `ssim` from `blox.torch.evaluation`.  PARITY UNPINNED: blox is absent (empty submodule), so this is the build's
written spec — mse over all elements of the [-1, 1] images; psnr = mean over frames of 10 log10(1 / mse_frame) on the
[0, 1]-scaled images; ssim = mean over frames and channels of skimage.metrics.structural_similarity's default map (7x7 uniform
window, K1 = 0.01, K2 = 0.03, data_range = 1, sample covariance, valid windows only).  The SSIM map itself IS pinned: scipy's
uniform_filter formulation that skimage uses is restated in `ssim_plane_reference` and must agree with the sliding-window sum."""
import numpy as np


def ssim_plane(a, b, win=7, K1=0.01, K2=0.03, data_range=1.0):
    """mean SSIM of two [H, W] planes (float64), direct sliding-window sums"""
    a, b = a.astype(np.float64), b.astype(np.float64)
    H, W = a.shape
    oh, ow = H - win + 1, W - win + 1
    def box(x):
        c = np.cumsum(np.cumsum(np.pad(x, ((1, 0), (1, 0))), 0), 1)
        return (c[win:, win:] - c[:-win, win:] - c[win:, :-win] + c[:-win, :-win])[:oh, :ow]
    n = win * win
    ux, uy = box(a) / n, box(b) / n
    cn = n / (n - 1.0)
    vx, vy, vxy = cn * (box(a * a) / n - ux * ux), cn * (box(b * b) / n - uy * uy), cn * (box(a * b) / n - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    return float(S.mean())


def ssim_plane_reference(a, b, win=7):
    """the same map the way skimage computes it (scipy.ndimage.uniform_filter, then cropping (win - 1) / 2 at every border)"""
    from scipy.ndimage import uniform_filter
    a, b = a.astype(np.float64), b.astype(np.float64)
    n = win * win
    cn = n / (n - 1.0)
    ux, uy = uniform_filter(a, win), uniform_filter(b, win)
    vx = cn * (uniform_filter(a * a, win) - ux * ux)
    vy = cn * (uniform_filter(b * b, win) - uy * uy)
    vxy = cn * (uniform_filter(a * b, win) - ux * uy)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    p = (win - 1) // 2
    return float(S[p:-p, p:-p].mean())


def sequence_metrics(gen, tgt):
    """gen, tgt [n, C, H, W] in [-1, 1] -> (mse, psnr, ssim) as Evaluator.compute_metrics stores them"""
    gen, tgt = np.asarray(gen, np.float64), np.asarray(tgt, np.float64)
    mse = float(np.mean((gen - tgt) ** 2))
    g01, t01 = (gen + 1) / 2, (tgt + 1) / 2
    per_frame = np.mean((g01 - t01) ** 2, axis=(1, 2, 3))
    psnr = float(np.mean(10.0 * np.log10(1.0 / np.maximum(per_frame, 1e-30))))
    ssim = float(np.mean([[ssim_plane(g01[f, c], t01[f, c]) for c in range(gen.shape[1])] for f in range(gen.shape[0])]))
    return mse, psnr, ssim
