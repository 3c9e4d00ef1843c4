"""ORACLE (test infrastructure only — never imported by the product path; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may use it).

numpy / torch-CPU restatement of the training-time paths of the auxiliary models (paths relative to /root/reference):

  InverseModel.sample_offsets / index_input          gcp/prediction/models/auxilliary_models/inverse_mdl.py:84-114
  InverseModel.forward (sampled pair) / .loss        inverse_mdl.py:136-191
  CostModel._general_cost / _fast_path_dist_cost     gcp/prediction/models/auxilliary_models/cost_mdl.py:81-117
  CostModel.forward / .loss                          cost_mdl.py:42-73
  EuclideanPathLength (the 25-room ground-truth cost, experiments/prediction/25room/gcp_tree/conf.py:35-37)
                                                     gcp/planning/cem/cost_fcn.py:9-21,49-54
  LengthPredictorModule sample (val_mode(pred_length=True))   base_gcp.py:219-226, auxilliary_models/misc.py:38-51

PINNED: the index draws (np.random order), the gathers and both ground-truth costs reproduce fixtures made by executing the
reference's own functions (tests/golden/make_ref_aux_goldens.py -> ref_aux_models.npz; tests/test_aux_models_oracle.py).
UNPINNED (blox, absent): `L2Loss` — this build's spec is mean(weights * (estimate - target)^2), the same reduction the
state-regression loss already uses — and `OneHotCategorical.sample`, whose torch RNG stream cannot be reproduced by a
kernel: the length draw is restated as inverse-CDF sampling of softmax(logits) from ONE uniform number per sequence, fed in.
"""
import numpy as np
import torch


def sample_inverse_offsets(end_ind, temp_dist=1, rng=np.random):
    """inverse_mdl.py:84-104 (take_first_tstep=False): per sequence t0 ~ U{0 .. end_ind - temp_dist}, then ONE vectorised
    draw of the temporal distances; same np.random call order as the reference."""
    end_ind = np.asarray(end_ind)
    bs = end_ind.shape[0]
    t0 = np.zeros(bs)
    for b in range(bs):
        assert end_ind[b] >= temp_dist
        t0[b] = rng.randint(0, end_ind[b] - temp_dist + 1, 1)[0]
    delta_t = rng.randint(1, temp_dist + 1, bs)
    t1 = t0 + delta_t
    return t0.astype(np.int64), t1.astype(np.int64)


def sample_cost_pairs(end_ind, rng=np.random):
    """cost_mdl.py:105-107: per sequence start ~ U{0 .. end_ind-1}, end ~ U{start+1 .. end_ind}, interleaved draws."""
    end_ind = np.asarray(end_ind)
    s, e = [], []
    for b in range(end_ind.shape[0]):
        si = rng.randint(0, end_ind[b], 1)[0]
        s.append(si)
        e.append(rng.randint(si + 1, end_ind[b] + 1, 1)[0])
    return np.asarray(s, dtype=np.int64), np.asarray(e, dtype=np.int64)


def euclidean_path_cost(traj, start_idx, end_idx):
    """CostModel._general_cost's ground truth with EuclideanPathLength(dense_cost=True) (cost_mdl.py:110-111,
    cost_fcn.py:14-21,49-54): the segment traj[b, s:e+1] with goal traj[b, e]; per step the L2 norm ALONG THE LAST AXIS of the
    difference to the next element (the appended goal makes the last step zero), summed over every other axis and over steps.
    For images [T,3,H,W] that is a norm over each pixel row, summed over channels and rows.  traj [B,T,...] float32 -> [B,1]."""
    traj = np.asarray(traj, dtype=np.float32)
    out = []
    for b in range(traj.shape[0]):
        seg = traj[b, start_idx[b]:end_idx[b] + 1]
        nxt = np.concatenate([seg[1:], traj[b, end_idx[b]][None]])
        out.append(np.sum(np.linalg.norm(nxt - seg, axis=-1)))
    return np.stack(out).astype(np.float32)[:, None]


def fast_path_pairs(end_ind, u0, u1):
    """cost_mdl.py:85-88 with the two torch.rand draws fed in (float32 arithmetic, then truncation)."""
    e = torch.as_tensor(end_ind).float()
    u0, u1 = torch.as_tensor(u0), torch.as_tensor(u1)
    s = u0 * (e - 1)
    t = u1 * (e - (s + 1)) + (s + 1)
    return s.long().numpy(), t.long().numpy()


def fast_path_cost(traj, start_idx, end_idx):
    """cost_mdl.py:94-98 for state sequences [B,T,D]: cumulative L2 step length between the two indices -> [B,1]."""
    traj = torch.as_tensor(traj)
    B = traj.shape[0]
    cum = torch.cumsum(torch.norm(traj[:, 1:] - traj[:, :-1], dim=-1), dim=1)
    cum = torch.cat((torch.zeros((B, 1), dtype=cum.dtype), cum), dim=1)
    ar = torch.arange(B)
    return (cum[ar, torch.as_tensor(end_idx)] - cum[ar, torch.as_tensor(start_idx)])[:, None].numpy()


def sample_length(logits, u):
    """base_gcp.py:222: end_ind = clamp(argmax(OneHotCategorical(logits).sample()), min=2), the categorical draw restated as the
    inverse CDF of softmax(logits) at u in [0, 1): the first index whose cumulative probability exceeds u (float64)."""
    p = torch.softmax(torch.as_tensor(logits).double(), dim=1)
    cdf = torch.cumsum(p, dim=1)
    u = torch.as_tensor(u).double()[:, None]
    idx = (cdf <= u).sum(1).clamp(max=p.shape[1] - 1)
    return torch.clamp(idx, min=2)


def l2_loss(estimates, targets, weights=1.0):
    """blox L2Loss (absent) — build spec: elementwise squared error times weights, mean over all elements."""
    return ((estimates - targets) ** 2 * weights).mean()
