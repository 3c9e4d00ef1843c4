"""Shared helpers for the parity tests (seeded synthetic inputs per BASELINE.md §2 / SURVEY.md §8d)."""
import numpy as np
import torch


from video_gcp_amd.synthetic import make_inputs  # noqa: F401,E402  (one generator for tests, bench and the trainer)


def maxdiff(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    d = (a - b).abs()
    return float(d.max()) if d.numel() else 0.0


def assert_close(a, b, atol, rtol=0.0, name=""):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.nonzero(bad.flatten())[0])
        raise AssertionError(f"{name}: {int(bad.sum())}/{bad.numel()} elements differ; max err {float(err.max()):.3e} "
                             f"(atol {atol}, rtol {rtol}); first bad flat index {i}: got {float(a.flatten()[i])} "
                             f"want {float(b.flatten()[i])}")
