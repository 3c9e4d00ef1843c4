"""The adaptive-binding oracle (oracle/adaptive_oracle.py) against goldens produced by EXECUTING the reference's own
soft_dtw / fast_gak / basic_dtw (tests/golden/make_ref_dtw_goldens.py), plus hand-checkable properties."""
import os

import numpy as np
import pytest
import torch

from oracle import adaptive_oracle as A

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dtw.npz"))


@pytest.mark.parametrize("i", range(int(G["sd_n"])))
def test_soft_dtw_matches_reference_goldens(i):
    cost, end = G[f"sd{i}_cost"], G[f"sd{i}_end"]
    B = cost.shape[0]
    fwd = A.fast_gak(-cost.astype(np.float64), np.zeros(B, dtype=np.int64))
    ref = G[f"sd{i}_fwd"]
    assert np.array_equal(np.isinf(fwd), np.isinf(ref))
    fin = ~np.isinf(ref)
    assert np.max(np.abs(fwd[fin] - ref[fin])) < 1e-12
    w = A.soft_dtw(cost, end)
    assert w.dtype == np.float32
    assert np.max(np.abs(w - G[f"sd{i}_w"])) < 1e-6


def test_soft_dtw_properties():
    rng = np.random.RandomState(1)
    cost = rng.rand(2, 15, 9).astype(np.float32)
    end = np.array([8, 4])
    w = A.soft_dtw(cost, end)
    # every node is aligned with exactly one frame: rows sum to one; frames after end_ind are never matched
    assert np.allclose(w.sum(2), 1.0, atol=1e-5)
    assert np.all(w[1, :, 5:] == 0)
    # first node matches frame 0, last node matches frame end_ind
    assert np.allclose(w[:, 0, 0], 1.0, atol=1e-6) and np.allclose(w[[0, 1], -1, end], 1.0, atol=1e-6)
    # every frame up to end_ind is covered by at least one node in expectation >= 1
    assert np.all(w[0].sum(0) >= 1 - 1e-5)
    wn = A.normalize(torch.from_numpy(w), 1)
    assert torch.allclose(wn[1].sum(0)[:5], torch.ones(5), atol=1e-5) and float(wn[1].sum(0)[5:].abs().max()) == 0.0


@pytest.mark.parametrize("i", range(int(G["bd_n"])))
def test_basic_dtw_matches_reference_goldens(i):
    d, D, (p, q) = A.basic_dtw(G[f"bd{i}_C"])
    assert abs(d - float(G[f"bd{i}_d"])) < 1e-12
    assert np.max(np.abs(D - G[f"bd{i}_D"])) < 1e-12
    assert np.array_equal(p, G[f"bd{i}_p0"]) and np.array_equal(q, G[f"bd{i}_p1"])


def test_batch_cdist_is_squared_l2():
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(2, 5, 3, 4, 4, generator=g), torch.randn(2, 7, 3, 4, 4, generator=g)
    want = ((a[:, :, None] - b[:, None]) ** 2).flatten(3).sum(-1)
    assert torch.allclose(A.batch_cdist(a, b, "sum"), want, atol=1e-4)
    assert torch.allclose(A.batch_cdist(a, b, "mean"), want / 48, atol=1e-5)


def test_dtw_matches_pick_path_cells():
    """DTWEvalBinding.get_single_matches: the chosen estimate of every target frame lies on the DTW path"""
    rng = np.random.RandomState(3)
    C = rng.rand(15, 9)
    d, D, path, inds = A.dtw_matches(C)
    cells = set(zip(path[0].tolist(), path[1].tolist()))
    assert all((int(inds[j]), j) in cells for j in range(9))
    assert inds[0] == 0 and np.all(np.diff(inds) >= 0)


# ---------------------------------------------------------------------------------------------------
# learned matching temperature (hyperparameters.py:132, adaptive.py:19-21, :51)
# ---------------------------------------------------------------------------------------------------
GT = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dtw_dtemp.npz"))


@pytest.mark.parametrize("i", range(int(GT["n"])))
def test_soft_dtw_autograd_values_and_temperature_derivative(i):
    """soft_dtw_autograd (a) returns soft_dtw's values, (b) differentiates w.r.t. the temperature like central differences of the
    reference's forward executed at temp +- h (the goldens), cell by cell; the reference's OWN autograd through soft_dtw is NaN for
    this parameter (stored in the goldens: the -inf cells of the lattice), so there is no finite reference gradient to match."""
    cost, end, t0, h = torch.from_numpy(GT[f"c{i}_cost"]), GT[f"c{i}_end"], float(GT[f"c{i}_temp"]), float(GT["h"])
    assert np.all(np.isnan(GT[f"c{i}_ref_autograd_dtemp"]))
    temp = torch.full((1,), t0, requires_grad=True)
    w = A.soft_dtw_autograd(cost / temp, end)
    w_np = A.soft_dtw((cost / t0).numpy(), end)
    assert np.max(np.abs(w.detach().numpy() - w_np)) < 1e-6 and np.max(np.abs(w_np - GT[f"c{i}_w"])) < 1e-6
    # forward-mode derivative of every cell at once: d w / d temp = jvp along temp
    _, dw = torch.autograd.functional.jvp(lambda t: A.soft_dtw_autograd(cost / t, end), temp.detach(), torch.ones(1))
    fd = (GT[f"c{i}_w_plus"].astype(np.float64) - GT[f"c{i}_w_minus"]) / (2 * h)
    scale = np.abs(fd).max()
    assert scale > 1e-3
    assert np.max(np.abs(dw.numpy() - fd)) < 5e-4 * scale + 5e-5
    (g,) = torch.autograd.grad((w * torch.from_numpy(GT[f"c{i}_G"])).sum(), temp)
    assert torch.isfinite(g).all()
    assert abs(float(g) - float((fd * GT[f"c{i}_G"]).sum())) < 3e-3 * float(np.abs(fd * GT[f"c{i}_G"]).sum())


def test_get_w_keeps_the_temperature_in_the_graph_only_when_it_is_learned():
    from video_gcp_amd.hparams import config
    hp_off, hp_on = config("c5s"), config("c5s", learn_matching_temp=True)
    g = torch.Generator().manual_seed(3)
    img, traj = torch.rand(2, 15, 3, 8, 8, generator=g), torch.rand(2, 12, 3, 8, 8, generator=g)
    end = torch.tensor([11, 6])
    temp = torch.full((1,), 0.7, requires_grad=True)
    sd = {"tree_module.tree_modules.0.binding.temp": temp}
    w_off, _ = A.get_w(hp_off, sd, img, traj, end)
    w_on, _ = A.get_w(hp_on, sd, img, traj, end)
    assert not w_off.requires_grad and w_on.requires_grad
    assert torch.allclose(w_off, w_on.detach(), atol=1e-6)
