"""CPU: pin oracle/tree_index_oracle.py with the hand-derived known answers of SURVEY.md App. A (the reference has
no tests of its own) and with fixtures produced by running the reference's own functions
(tests/golden/make_ref_goldens.py -> tests/golden/ref_tree_utils.npz)."""
import os

import numpy as np
import pytest

from oracle import tree_index_oracle as TI

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_kat1_depthfirst2layers():
    # tree_utils.py:222-232
    assert [l.tolist() for l in TI.depthfirst2layers(np.arange(7))] == [[3], [1, 5], [0, 2, 4, 6]]
    got = [l.tolist() for l in TI.depthfirst2layers(np.arange(15))]
    assert got == [[7], [3, 11], [1, 5, 9, 13], [0, 2, 4, 6, 8, 10, 12, 14]]
    assert TI.depthfirst2breadthfirst(np.arange(7)).tolist() == [3, 1, 5, 0, 2, 4, 6]


def test_kat2_interleave():
    a, b = np.array([[10, 11, 12]]), np.array([[20, 21, 22]])
    assert TI.interleave(a, b).tolist() == [[10, 20, 11, 21, 12, 22]]


def test_kat3_bf_df_maps():
    for L in range(1, 9):
        perm = TI.bf2df_perm(L)
        assert sorted(perm.tolist()) == list(range(2 ** L - 1))
        # bf order = concatenated layers of the df sequence
        assert np.array_equal(TI.depthfirst2breadthfirst(np.arange(2 ** L - 1)), perm)
    assert TI.df_index(0, 0, 3) == 3 and TI.bf_index(2, 3) == 6 and TI.df_index(2, 3, 3) == 6


def test_kat4a_balanced_L3_end4():
    # SURVEY App. A.4 worked example: df timesteps [0p,0,1,2,2p,3,4]
    ts, cs = TI.balanced_layers([4], 3, 7)
    assert ts[0].tolist() == [[2]] and ts[1].tolist() == [[0, 3]] and ts[2].tolist() == [[0, 1, 2, 4]]
    leave = TI.leave_mask_df([4], 3, 7)[0]
    assert leave.tolist() == [False, True, True, True, False, True, True]
    perm = TI.bf2df_perm(3)
    t_df = np.zeros(7, dtype=int)
    t_df[perm] = TI.balanced_timesteps_bf([4], 3, 7)[0]
    assert t_df.tolist() == [0, 0, 1, 2, 2, 3, 4]
    assert t_df[leave].tolist() == [0, 1, 2, 3, 4]


@pytest.mark.parametrize("L,end", [(3, 6), (5, 19), (7, 2), (7, 40), (7, 79)])
def test_kat4_selected(L, end):
    T = 2 ** L
    leave = TI.leave_mask_df([end], L, T)[0]
    perm = TI.bf2df_perm(L)
    t_df = np.zeros(2 ** L - 1, dtype=int)
    t_df[perm] = TI.balanced_timesteps_bf([end], L, T)[0]
    assert t_df[leave].tolist() == list(range(end + 1))
    bf = TI.brute_force_kept_timesteps(end, L)
    assert [t for _, t, _ in bf] == t_df.tolist()
    assert [k for _, _, k in bf] == leave.tolist()


def test_kat4_property_exhaustive():
    """kept depth-first timesteps are exactly 0..end for every L in 2..8 and 0 <= end <= 2^L - 2."""
    for L in range(2, 9):
        ends = np.arange(0, 2 ** L - 1)
        T = 2 ** L
        leave = TI.leave_mask_df(ends, L, T)
        perm = TI.bf2df_perm(L)
        t_df = np.zeros((len(ends), 2 ** L - 1), dtype=int)
        t_df[:, perm] = TI.balanced_timesteps_bf(ends, L, T)
        md = TI.balanced_match_dist(ends, L, T)
        for i, e in enumerate(ends):
            assert t_df[i][leave[i]].tolist() == list(range(e + 1))
            # every valid frame column of match_dist is one-hot, padded columns are all zero (argmax -> 0, D5)
            assert np.array_equal(md[i].sum(0), (np.arange(T) <= e).astype(np.float32))
        idx = TI.matched_node_index(md)
        assert np.all(idx[np.arange(T)[None] > ends[:, None]] == 0)


def test_kat5_pad_mask():
    assert TI.get_pad_mask(np.array([0, 2, 4]), 5).tolist() == [[1, 0, 0, 0, 0], [1, 1, 1, 0, 0], [1, 1, 1, 1, 1]]


def test_torch13_vs_true_division_divergence():
    """F4: with true division (torch >= 1.7) the L=3,end=4 tree would keep duplicates; the oracle must not."""
    leave = TI.leave_mask_df([4], 3, 7)[0]
    assert leave.sum() == 5


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "ref_tree_utils.npz")), reason="fixture missing")
def test_against_reference_generated_goldens():
    g = np.load(os.path.join(GOLD, "ref_tree_utils.npz"))
    for n in (7, 15, 127):
        x = g[f"d2l_in_{n}"]
        layers = TI.depthfirst2layers(x, axis=1)
        for i, l in enumerate(layers):
            assert np.array_equal(l, g[f"d2l_out_{n}_{i}"])
        assert np.array_equal(TI.depthfirst2breadthfirst(x, axis=1), g[f"d2b_out_{n}"])
    assert np.array_equal(TI.interleave(g["il_a"], g["il_b"]), g["il_out"])
    assert np.array_equal(TI.get_pad_mask(g["pm_end"], 20), g["pm_out"])
