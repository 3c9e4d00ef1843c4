"""ORACLE (test infrastructure only — never imported by the product path).

Plain-numpy / pure-Python restatement of the reference's *integer* tree bookkeeping for gcp_tree.
Every function cites the reference file:line it follows (paths relative to /root/reference).

Pinning status: the reference ships no tests/golden vectors.  These functions are pinned
by (a) the hand-derived known answers of SURVEY.md App. A (tests/test_tree_index_oracle.py) and
(b) fixtures produced by executing the reference's own `tree_utils.depthfirst2layers/interleave`,
`tree_optimizer` and `dtw_utils.basic_dtw` in the build container (tests/golden/make_ref_goldens.py).
The balanced-binding midpoint follows the torch-1.3 Long/Long rule the reference pins
(requirements.txt:18): C-style truncation toward zero.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------
# tree_utils.py:222-232  depthfirst2layers
# ------------------------------------------------------------------------------------------------
def depthfirst2layers(x, axis=0):
    """Split a depth-first (in-order) sequence of 2^d-1 items into layers, root layer first."""
    x = np.asarray(x)
    n = x.shape[axis]
    depth = int(np.log2(n + 1))
    assert 2 ** depth - 1 == n
    slices = []
    for _ in range(depth):
        idx_even = [slice(None)] * x.ndim
        idx_even[axis] = slice(0, None, 2)
        idx_odd = [slice(None)] * x.ndim
        idx_odd[axis] = slice(1, None, 2)
        slices.append(x[tuple(idx_even)])
        x = x[tuple(idx_odd)]
    return list(reversed(slices))


def depthfirst2breadthfirst(x, axis=0):
    """tree_utils.py:217-219."""
    return np.concatenate(depthfirst2layers(x, axis), axis)


def interleave(a, b):
    """tree_utils.py:202-205: out[:, 2k] = a[:, k], out[:, 2k+1] = b[:, k] (dim 1)."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    return np.stack((a, b), axis=2).reshape(a.shape[0], 2 * a.shape[1], *a.shape[2:])


def bf_index(level, j):
    """SURVEY App. A.3 (from tree_utils.py:101-108): node j of level `level` in breadth-first order."""
    return 2 ** level - 1 + j


def df_index(level, j, depth):
    """SURVEY App. A.3 (from tree_utils.py:79-88): in-order position of node j of level `level`."""
    return (2 * j + 1) * 2 ** (depth - 1 - level) - 1


def bf2df_perm(depth):
    """perm[bf] = df for a full tree of `depth` levels."""
    perm = np.zeros(2 ** depth - 1, dtype=np.int64)
    for l in range(depth):
        for j in range(2 ** l):
            perm[bf_index(l, j)] = df_index(l, j, depth)
    return perm


# ------------------------------------------------------------------------------------------------
# frame_binding.py:37-65  BalancedBinding (integer part), tree_utils.py:50-68 apply_fn
# ------------------------------------------------------------------------------------------------
def _trunc_div2(s):
    """torch-1.3 Long / 2: C truncation toward zero."""
    s = np.asarray(s, dtype=np.int64)
    return np.where(s >= 0, s // 2, -((-s) // 2))


def balanced_layers(end_ind, depth, max_seq_len):
    """Walk the tree level by level exactly like `BaseBinding.apply_tree` -> `SubgoalTreeLayer.apply_fn`
    -> `BalancedBinding.__call__` (frame_binding.py:22-26, 42-50; tree_utils.py:50-68).

    Returns per level: timesteps int64 [B, 2^l] and c_n_prime float32 [B, 2^l, T].
    """
    end_ind = np.asarray(end_ind, dtype=np.int64)
    B = end_ind.shape[0]
    left = np.zeros((B, 1), dtype=np.int64) - 1           # frame_binding.py:62-65
    right = end_ind[:, None] + 1
    ts_layers, c_layers = [], []
    for _ in range(depth):
        t = _trunc_div2(left + right)                      # frame_binding.py:52-54
        c = np.zeros(t.shape + (max_seq_len,), dtype=np.float32)
        bb, nn = np.meshgrid(np.arange(B), np.arange(t.shape[1]), indexing="ij")
        c[bb, nn, t] = 1.0                                 # make_one_hot, frame_binding.py:44
        c[left == t] = 0                                   # frame_binding.py:47
        c[right == t] = 0                                  # frame_binding.py:48
        ts_layers.append(t)
        c_layers.append(c)
        left, right = interleave(left, t), interleave(t, right)   # tree_utils.py:65-68
    return ts_layers, c_layers


def balanced_match_dist(end_ind, depth, max_seq_len):
    """match_dist = tree.bf.c_n_prime, float32 [B, N, T] in bf order (frame_binding.py:56-60)."""
    _, c = balanced_layers(end_ind, depth, max_seq_len)
    return np.concatenate(c, axis=1)


def balanced_timesteps_bf(end_ind, depth, max_seq_len):
    t, _ = balanced_layers(end_ind, depth, max_seq_len)
    return np.concatenate(t, axis=1)


def matched_node_index(match_dist):
    """frame_binding.py:30: indices = match_dist.argmax(1) -> bf node index per frame, [B, T].
    All-zero columns (padded frames) give 0."""
    return np.argmax(match_dist, axis=1).astype(np.int64)


def leave_mask_df(end_ind, depth, max_seq_len):
    """evaluation_matching.py:201-204: `leave` = c_n_prime.any(-1) in depth-first order, bool [B, N]."""
    md = balanced_match_dist(end_ind, depth, max_seq_len)            # bf
    perm = bf2df_perm(depth)
    leave_bf = md.any(-1)
    leave_df = np.zeros_like(leave_bf)
    leave_df[:, perm] = leave_bf
    return leave_df


def get_pad_mask(end_ind, max_seq_len):
    """gcp/prediction/utils/utils.py:30-50."""
    end_ind = np.asarray(end_ind)
    return (np.arange(max_seq_len) <= end_ind[:, None]).astype(np.float32)


def brute_force_kept_timesteps(end, depth):
    """Independent recursive restatement used to cross-check `balanced_layers` (tree_module.py:90-91 via
    tree_utils.py:21-44 recursion).  Returns [(df_pos, t, kept)] for one sample."""
    out = []

    def rec(tl, tr, lo, hi):           # node occupies df positions (lo, hi) exclusive midpoint
        if hi - lo < 2:
            return
        pos = (lo + hi) // 2
        s = tl + tr
        t = s // 2 if s >= 0 else -((-s) // 2)
        rec(tl, t, lo, pos)
        out.append((pos, t, not (t == tl or t == tr)))
        rec(t, tr, pos, hi)

    n = 2 ** depth - 1
    rec(-1, end + 1, -1, n)
    return sorted(out)
