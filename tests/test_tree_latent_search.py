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
