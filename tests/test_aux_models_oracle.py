"""The auxiliary-model oracle against fixtures produced by EXECUTING the reference's functions
(tests/golden/make_ref_aux_goldens.py: InverseModel.sample_offsets / index_input, CostModel._general_cost with
EuclideanPathLength, CostModel._fast_path_dist_cost).  Index draws and gathers are bit-exact; the float32 path costs are
compared at 1e-5 relative (numpy's pairwise float32 sum inside the reference vs. the oracle's own summation order)."""
import os

import numpy as np
import torch

from oracle import aux_models_oracle as AX

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_aux_models.npz"))


def test_inverse_model_offsets_reproduce_reference_draws():
    for case in range(3):
        temp_dist, seed = G[f"inv{case}_cfg"]
        t0, t1 = AX.sample_inverse_offsets(G[f"inv{case}_end_ind"], int(temp_dist), np.random.RandomState(int(seed)))
        assert np.array_equal(t0, G[f"inv{case}_t0"]) and np.array_equal(t1, G[f"inv{case}_t1"])
        B = len(t0)
        ar = np.arange(B)
        assert np.array_equal(G[f"inv{case}_actions"][ar, t0], G[f"inv{case}_sel_actions"])
        assert np.array_equal(G[f"inv{case}_enc"][ar, t1], G[f"inv{case}_sel_enc"])
        assert (t1 <= G[f"inv{case}_end_ind"]).all() and (t0 >= 0).all()


def test_cost_pairs_and_euclidean_path_cost():
    for case in range(3):
        e = G[f"cost{case}_end_ind"]
        s_idx, e_idx = AX.sample_cost_pairs(e, np.random.RandomState(int(G[f"cost{case}_seed"][0])))
        assert np.array_equal(s_idx, G[f"cost{case}_start_idx"]) and np.array_equal(e_idx, G[f"cost{case}_end_idx"])
        ar = np.arange(len(e))
        assert np.array_equal(G[f"cost{case}_mes"][ar, s_idx], G[f"cost{case}_start"])
        assert np.array_equal(G[f"cost{case}_mes"][ar, e_idx], G[f"cost{case}_end"])
        gt = AX.euclidean_path_cost(G[f"cost{case}_traj"], s_idx, e_idx)
        assert gt.shape == G[f"cost{case}_gt"].shape
        np.testing.assert_allclose(gt, G[f"cost{case}_gt"], rtol=1e-5)


def test_fast_path_dist_cost():
    for case in range(2):
        e = G[f"fast{case}_end_ind"]
        s_idx, e_idx = AX.fast_path_pairs(e, G[f"fast{case}_u0"], G[f"fast{case}_u1"])
        ar = np.arange(len(e))
        assert np.array_equal(G[f"fast{case}_mes"][ar, s_idx], G[f"fast{case}_start"])
        assert np.array_equal(G[f"fast{case}_mes"][ar, e_idx], G[f"fast{case}_end"])
        np.testing.assert_allclose(AX.fast_path_cost(G[f"fast{case}_traj"], s_idx, e_idx), G[f"fast{case}_gt"], rtol=1e-6, atol=1e-6)


def test_sample_length_inverse_cdf():
    logits = torch.tensor([[0.0, 0.0, 0.0, 0.0], [10.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 10.0], [0.0, 0.0, 5.0, 5.0]])
    got = AX.sample_length(logits, torch.tensor([0.80, 0.5, 0.5, 0.25]))
    # uniform: cdf = .25 .5 .75 1 -> u=.8 lands in bin 3; a draw of bin 0 is clamped to 2 (base_gcp.py:222); bin 3; bin 2
    assert got.tolist() == [3, 2, 3, 2]
    assert AX.sample_length(logits[:1], torch.tensor([0.999999999])).tolist() == [3]
