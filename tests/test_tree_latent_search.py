"""CPU: the hierarchical latent optimizer against golden vectors produced by EXECUTING the reference class
(tests/golden/make_ref_planner_goldens.py -> ref_tree_optimizer.npz): samples (RNG draw order), selected plans, costs and
completion flags must match bit for bit."""
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_matches_reference_goldens(case):
    from planner_stubs import StubCost, stub_rollouts
    from video_gcp_amd.tree_latent_search import ImageHierarchicalTreeLatentOptimizer
    g = np.load(os.path.join(GOLD, "ref_tree_optimizer.npz"))
    cfg = g[f"c{case}_cfg"].tolist()
    depth, n_ll, ld, seed, rates = cfg[0], cfg[1], cfg[2], cfg[3], cfg[4:]
    np.random.seed(seed)
    cost = StubCost()
    opt = ImageHierarchicalTreeLatentOptimizer(ld, list(rates), depth, cost, cost, n_ll)
    goal = np.random.rand(1, 2, 2, 3)
    assert np.array_equal(goal, g[f"c{case}_goal"])
    for it in range(len(rates) + 1):
        z = opt.sample()
        assert np.array_equal(z, g[f"c{case}_it{it}_z"]), f"samples differ at iteration {it}"
        best, c = opt.optimize(stub_rollouts(z), goal)
        assert np.array_equal(np.asarray(best), g[f"c{case}_it{it}_best_rollout"])
        assert np.array_equal(np.asarray(c, dtype=np.float64).reshape(-1), g[f"c{case}_it{it}_best_cost"])
        assert bool(opt.fully_optimized) == bool(g[f"c{case}_it{it}_fully"][0])
    assert np.array_equal(opt.sample(), g[f"c{case}_final_z"])


def test_sample_layout_is_depth_first():
    """batch sizes 10, 10, 5 for depth 8, rates [10, 10], n_ll 5; [left | node | right]."""
    from planner_stubs import StubCost
    from video_gcp_amd.tree_latent_search import HierarchicalTreeLatentOptimizer
    np.random.seed(0)
    opt = HierarchicalTreeLatentOptimizer(3, [10, 10], 8, StubCost(), StubCost(), 5)
    z = opt.sample()
    assert z.shape == (10, 255, 3)
    # the root latent sits in the middle of the depth-first axis and differs per sample
    assert len({tuple(z[i, 127]) for i in range(10)}) == 10


def test_generator_draws_run_the_same_search_on_fewer_gaussians():
    """rng=np.random.Generator: only the rows the search keeps are drawn (below the level being optimised the reference draws
    n_samples rows and keeps the first, tree_optimizer.py:76-82).  Same shapes, layout, completion schedule and selection rule as the
    reference-stream mode; reproducible from the seed; the module-level np.random stream is left untouched."""
    from planner_stubs import StubCost, stub_rollouts
    from video_gcp_amd.tree_latent_search import ImageHierarchicalTreeLatentOptimizer

    def search(rng, seed=5):
        np.random.seed(seed)
        cost = StubCost()
        opt = ImageHierarchicalTreeLatentOptimizer(4, [3, 2], 5, cost, cost, 4, rng=rng)
        goal = np.random.rand(1, 2, 2, 3)
        trace = []
        for it in range(3):
            z = opt.sample()
            best, c = opt.optimize(stub_rollouts(z), goal)
            trace.append((z, np.asarray(best), bool(opt.fully_optimized)))
        return trace, opt.sample()
    ref_trace, ref_final = search(None)
    a_trace, a_final = search(np.random.default_rng(11))
    after = np.random.get_state()[1].copy()
    np.random.seed(5)
    np.random.rand(1, 2, 2, 3)                                                   # (the goal image of `search` is the only legacy draw)
    assert np.array_equal(np.random.get_state()[1], after), "generator mode must not consume the legacy stream"
    b_trace, b_final = search(np.random.default_rng(11))
    for (za, ba, fa), (zb, bb, fb), (zr, br, fr) in zip(a_trace, b_trace, ref_trace):
        assert np.array_equal(za, zb) and np.array_equal(ba, bb)                  # reproducible
        assert za.shape == zr.shape and za.dtype == np.float32 and fa == fr       # same populations, same completion schedule
    assert a_final.shape == ref_final.shape == (1, 31, 4) and np.array_equal(a_final, b_final)
    # what was fixed in iteration 0 (the root latent of the chosen sample) stays in every later population
    root = a_trace[1][0][:, 15]
    assert all(np.array_equal(root[0], r) for r in root) and any(np.array_equal(root[0], z15) for z15 in a_trace[0][0][:, 15])
    # the draws are standard normal
    big = ImageHierarchicalTreeLatentOptimizer(64, [10, 10], 7, StubCost(), StubCost(), 5, rng=np.random.default_rng(0)).sample()
    assert big.shape == (10, 127, 64) and abs(float(big.mean())) < 0.02 and abs(float(big.std()) - 1.0) < 0.02


def test_torch_generator_draws_have_the_same_layout():
    """rng=torch.Generator (what the device-resident planner uses): `sample()` is built on the generator's device — same shape, same
    depth-first layout ([left | node | right]: a fixed root latent sits in the middle of every row), reproducible from the seed"""
    import torch
    from planner_stubs import StubCost
    from video_gcp_amd.tree_latent_search import ImageHierarchicalTreeLatentOptimizer

    def make(seed):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        return ImageHierarchicalTreeLatentOptimizer(8, [4, 3], 5, StubCost(), StubCost(), 2, rng=g)
    a, b = make(1), make(1)
    za, zb = a.sample(), b.sample()
    assert torch.is_tensor(za) and za.shape == (4, 31, 8) and za.dtype == torch.float32 and torch.equal(za, zb)
    assert len({tuple(za[i, 15].tolist()) for i in range(4)}) == 4                 # one root latent per sample
    # fix the root the way _opt_subgoal does and draw again: every row carries it, the level below is now the sampled one
    root = a._root
    root.best_z, root.done, root.n_samples = root.last_draw[2], True, 1
    root.left, root.right = root.left[:1], root.right[:1]
    z2 = a.sample()
    assert z2.shape == (3, 31, 8) and all(torch.equal(z2[i, 15], za[2, 15]) for i in range(3))
    assert len({tuple(z2[i, 7].tolist()) for i in range(3)}) == 3                  # left child's subgoal latent: one per sample
    big = make(0)
    big._dim = 64
    assert abs(float(torch.cat([big.sample() for _ in range(20)]).std()) - 1.0) < 0.02
