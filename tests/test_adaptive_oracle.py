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
