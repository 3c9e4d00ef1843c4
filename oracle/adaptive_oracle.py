"""ORACLE (test infrastructure only — never imported by the product path; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).

CPU restatement of the adaptive-binding path of the reference (config c5: `matching_type='dtw_image'`,
`attentive_inference=True`, experiments/prediction/base_configs/gcp_adaptive.py:6-11).  Paths relative to
/root/reference:

  fast_gak / soft_dtw                       gcp/prediction/models/adaptive_binding/probabilistic_dtw.py:11-73, 82-122
  AdaptiveBinding.get_w / prune_sequence    gcp/prediction/models/adaptive_binding/adaptive.py:32-77
  LossAveragingCriterion.loss / soft avg    gcp/prediction/models/adaptive_binding/binding_loss.py:19-58
  AttentiveInference / Attention.forward    gcp/prediction/models/adaptive_binding/attentive_inference.py:11-86
  basic_dtw / _traceback                    gcp/evaluation/dtw_utils.py:77-95, 201-218 (what evaluation_matching.py:12-15 imports)

PINNING: fast_gak / soft_dtw / basic_dtw are pinned by goldens produced by EXECUTING the reference
functions in the build container (tests/golden/make_ref_dtw_goldens.py -> tests/golden/ref_dtw.npz).
PARITY UNPINNED at the `blox` boundary for: `batch_cdist`, `normalize`, `safe_entropy`, `MultiheadAttention`,
`AttnKeyEncodingModule` (un-vendored submodule, absent).  Those follow THIS build's written spec, stated in each
docstring below and in DESIGN.md.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
# soft-DTW (expected edge frequencies of the 'nohor' alignment lattice), float64
# ---------------------------------------------------------------------------------------------------
def _lse2(a, b):
    """torch.logsumexp over a stacked pair (probabilistic_dtw.py:59): max + log(exp(a-max) + exp(b-max)), -inf safe."""
    m = np.maximum(a, b)
    m0 = np.where(np.isinf(m), 0.0, m)
    with np.errstate(divide="ignore"):
        return np.log(np.exp(a - m0) + np.exp(b - m0)) + m0


def fast_gak(C, begin_inds):
    """probabilistic_dtw.py:11-73, transition='nohor'.  C float64 [B, r, c] (log-domain scores), r >= c.
    D[b, 0, begin] = C[b, 0, begin]; D[i, j] = C[i, j] + logsumexp(D[i-1, j], D[i-1, j-1]) swept over anti-diagonals;
    a cell that is already finite (the begin cell) is left as is.  Indices -1 wrap to the last row / column exactly
    like the reference's fancy indexing (they only ever read -inf)."""
    C = np.asarray(C, dtype=np.float64)
    B, r, c = C.shape
    assert r >= c
    D = np.full_like(C, -np.inf)
    ar = np.arange(B)
    D[ar, 0, begin_inds] = C[ar, 0, begin_inds]
    for i in range(1, r + c):
        jds = np.arange(i + 1)[max(0, i - r + 1):c]
        ids = i - jds
        skip = D[:, ids - 1, jds]
        step = D[:, ids - 1, jds - 1]
        new = C[:, ids, jds] + _lse2(skip, step)
        old = D[:, ids, jds]
        D[:, ids, jds] = np.where(old != -np.inf, old, new)
    return D


def soft_dtw(cost, end_inds=None):
    """probabilistic_dtw.py:82-122.  cost [B, r, c] (any float dtype) -> w float32 [B, r, c]: probability that node i
    is aligned with frame j under the Gibbs distribution over monotone 'nohor' alignments that start at (0, 0) and end
    at (r-1, end_ind)."""
    C = -np.asarray(cost, dtype=np.float64)
    B, r, c = C.shape
    end = np.full(B, c - 1, dtype=np.int64) if end_inds is None else np.asarray(end_inds, dtype=np.int64)
    comb = np.concatenate([C, C[:, ::-1, ::-1]], 0)                         # :101
    begin = np.concatenate([np.zeros_like(end), c - end - 1], 0)            # :103
    acc = fast_gak(comb, begin)
    fwd, bwd = acc[:B], acc[B:, ::-1, ::-1]
    z = fwd[np.arange(B), -1, end][:, None, None]                           # :111
    e = fwd + bwd - C
    e[C == -np.inf] = -np.inf
    with np.errstate(invalid="ignore"):
        w = np.exp(e - z)
    return w.astype(np.float32)


def soft_dtw_autograd(cost, end_inds):
    """soft_dtw (probabilistic_dtw.py:82-122) as torch float64 ops, for a cost that carries a gradient (the learned matching
    temperature, adaptive.py:19-21, :51).  Same cell recurrence as fast_gak above, swept row by row instead of along anti-diagonals
    (a 'nohor' cell reads the previous row only, so every cell sees the same two operands); test_adaptive_cpu checks its values
    against soft_dtw() bit for bit where both apply.  cost [B, r, c] torch -> w float32 [B, r, c] with grad."""
    C = (-cost).double()
    B, r, c = C.shape
    end = torch.as_tensor(end_inds, dtype=torch.long)
    ninf = torch.full((B, 1), -np.inf, dtype=torch.float64)

    def sweep(Cm, begin):
        first = torch.full((B, c), -np.inf, dtype=torch.float64)
        first = first.scatter(1, begin[:, None], Cm[:, 0].gather(1, begin[:, None]))
        rows = [first]
        for i in range(1, r):
            skip = rows[-1]
            step = torch.cat([ninf, skip[:, :-1]], 1)
            m = torch.maximum(skip, step)
            dead = torch.isinf(m)                  # neither predecessor is reachable: the cell is -inf and carries no derivative
            zero = torch.zeros_like(m)             # (the reference's logsumexp gives such cells a NaN derivative, see the goldens' header)
            sk, st, m0 = torch.where(dead, zero, skip), torch.where(dead, zero, step), torch.where(dead, zero, m)
            lse = torch.log(torch.exp(sk - m0) + torch.exp(st - m0)) + m0
            rows.append(torch.where(dead, m, Cm[:, i] + lse))
        return torch.stack(rows, 1)

    fwd = sweep(C, torch.zeros_like(end))
    bwd = sweep(C.flip(1, 2), c - end - 1).flip(1, 2)
    z = fwd[torch.arange(B), -1, end][:, None, None]
    e = fwd + bwd - C
    # cells no alignment passes through are -inf in one of the two sweeps: w = 0 and no gradient
    reach = torch.isfinite(e)
    w = torch.where(reach, torch.exp(torch.where(reach, e, torch.zeros_like(e)) - z), torch.zeros_like(e))
    return w.float()


# ---------------------------------------------------------------------------------------------------
# hard DTW of the evaluation harness (dtw_utils.py)
# ---------------------------------------------------------------------------------------------------
def _traceback(D):
    """dtw_utils.py:201-218 on the (r+1, c+1) padded accumulator."""
    i, j = D.shape[0] - 2, D.shape[1] - 2
    p, q = [i], [j]
    while i > 0 or j > 0:
        tb = int(np.argmin((D[i, j], D[i, j + 1], D[i + 1, j])))
        if tb == 0:
            i, j = i - 1, j - 1
        elif tb == 1:
            i -= 1
        else:
            j -= 1
        p.insert(0, i)
        q.insert(0, j)
    return np.array(p), np.array(q)


def basic_dtw(C):
    """dtw_utils.py:77-95: (normalised distance, accumulated cost [r, c], path)."""
    r, c = C.shape
    D = np.zeros((r + 1, c + 1))
    D[0, 1:] = np.inf
    D[1:, 0] = np.inf
    D[1:, 1:] = C
    for i in range(r):
        for j in range(c):
            D[i + 1, j + 1] += min(D[i, j], D[i + 1, j], D[i, j + 1])
    return D[-1, -1] / (r + c), D[1:, 1:], _traceback(D)


def dtw_matches(cost):
    """DTWEvalBinding.get_single_matches (evaluation_matching.py:133-146) after the cdist: cost [n_estimates, n_targets] ->
    (normalised distance, accumulated cost, path, for every target the estimate chosen)."""
    d, D, path = basic_dtw(np.asarray(cost))
    match = np.full_like(D, np.inf)
    match[path[0], path[1]] = D[path[0], path[1]]
    return d, D, path, np.argmin(match, axis=0)


# ---------------------------------------------------------------------------------------------------
# blox-side helpers: THIS build's spec
# ---------------------------------------------------------------------------------------------------
def batch_cdist(x1, x2, reduction="sum"):
    """blox.torch.ops.batch_cdist (adaptive.py:44, binding_loss.py:24) — spec: squared L2 distance between every pair
    of flattened vectors by the quadratic expansion |a|^2 + |b|^2 - 2 a.b, clamped at 0; 'mean' divides by the vector
    length.  x1 [B, n, ...], x2 [B, m, ...] -> [B, n, m]."""
    a, b = x1.flatten(2), x2.flatten(2)
    an, bn = a.pow(2).sum(-1, keepdim=True), b.pow(2).sum(-1, keepdim=True)
    res = (an + bn.transpose(1, 2) - 2.0 * torch.bmm(a, b.transpose(1, 2))).clamp_min(0.0)
    if reduction == "mean":
        res = res / a.shape[2]
    return res


def normalize(w, dim):
    """blox.torch.dist.normalize (adaptive.py:58) — spec: w / max(sum over dim, 1e-7) (columns of padded frames are all
    zero and stay zero)."""
    return w / w.sum(dim, keepdim=True).clamp_min(1e-7)


def safe_entropy(p, dim=-1):
    """blox.torch.dist.safe_entropy (tree_module.py:145) — spec: -sum p log p with 0 log 0 = 0."""
    return -(torch.where(p > 0, p * torch.log(p.clamp_min(1e-30)), torch.zeros_like(p))).sum(dim)


def multihead_attention(sd, prefix, hp, query, keys, values, s_ind, e_ind):
    """blox MultiheadAttention(hp) called as attention(query, keys, values, s_ind, e_ind)
    (attentive_inference.py:81) — spec:
      q = q_proj(query) [R, dk]; k = k_proj(keys) [R, T, dk]; v = v_proj(values) [R, T, nz]; h heads split dk and nz;
      score[r, h, t] = q_h . k_h[t] / sqrt(dk / h) / temperature; frames outside [s_ind, e_ind] get -inf; softmax over t;
      out = out_proj(concat_h sum_t a[r, h, t] v_h[t]); returned weights = mean over heads."""
    h, dk, nz = hp.n_attention_heads, hp.nz_attn_key, hp.nz_enc
    lin = lambda x, n: F.linear(x, sd[f"{prefix}.{n}.weight"], sd[f"{prefix}.{n}.bias"])
    q, k, v = lin(query, "q_proj"), lin(keys, "k_proj"), lin(values, "v_proj")
    R, T = k.shape[:2]
    q = q.view(R, h, dk // h)
    k = k.view(R, T, h, dk // h)
    v = v.view(R, T, h, nz // h)
    score = torch.einsum("rhd,rthd->rht", q, k) / math.sqrt(dk // h) / sd[f"{prefix}.temperature"]
    t = torch.arange(T)[None, None, :]
    outside = (t < s_ind[:, None, None]) | (t > e_ind[:, None, None])
    a = torch.softmax(score.masked_fill(outside, float("-inf")), dim=-1)
    o = torch.einsum("rht,rthd->rhd", a, v).reshape(R, nz)
    return lin(o, "out_proj"), a.mean(1)


def attention(sd, prefix, hp, predictor, values, keys, query_input, start_ind, end_ind):
    """Attention.forward (attentive_inference.py:47-86) with mask_inf_attention=False (hyperparameters.py:126): the mask
    is the sequence's own [start_ind, end_ind].  values [B, T, nz], keys [B, T, dk]; query rows R = B * mult (b-major)."""
    query = predictor(sd, f"{prefix}.query_net", hp, *query_input)                       # :67
    mult = query.shape[0] // keys.shape[0]
    tile = lambda x: x.repeat_interleave(mult, 0)                                       # :73
    values, keys, s_ind, e_ind = tile(values), tile(keys), tile(start_ind), tile(end_ind)
    raw, att = None, None
    for i in range(hp.n_attention_layers):                                              # :80-83
        raw, att = multihead_attention(sd, f"{prefix}.attention_layers.{i}", hp, query, keys, values, s_ind, e_ind)
        x = F.layer_norm(raw, raw.shape[1:])
        query = F.layer_norm(predictor(sd, f"{prefix}.predictor_layers.{i}", hp, x) + query, query.shape[1:])
    return F.linear(raw, sd[f"{prefix}.out.weight"], sd[f"{prefix}.out.bias"]), att     # :85


def attn_key_encoder(sd, hp, enc_traj_seq, seq_encoder, training):
    """inf_key_encoder = Sequential(ConvSeqEncodingModule, AttnKeyEncodingModule(add_time=False)) (base_gcp.py:122-123)
    — spec of the absent AttnKeyEncodingModule: one Linear(nz_enc -> nz_attn_key) per frame."""
    x = seq_encoder(sd, hp, enc_traj_seq, training, prefix="inf_key_encoder.0")
    return F.linear(x, sd["inf_key_encoder.1.linear.weight"], sd["inf_key_encoder.1.linear.bias"])


def hack_weights_df(w, hp):
    """WeightsHacker.hack_weights_df (binding_loss.py:76-87): identity for the default top_bias = 1."""
    assert hp.top_bias == 1.0 and hp.leaves_bias == 0.0
    return w


def get_w(hp, sd, images_df, traj_seq, end_ind):
    """AdaptiveBinding.get_w (adaptive.py:32-60) up to (not including) depthfirst2breadthfirst: depth-first w [B, N, T]
    and the depth-first image cost matrix."""
    assert hp.matching_type == "dtw_image"
    cost = hack_weights_df(batch_cdist(images_df, traj_seq, reduction="mean"), hp)
    temp = sd["tree_module.tree_modules.0.binding.temp"]
    if hp.learn_matching_temp and temp.requires_grad:                    # adaptive.py:19-21: the division by temp stays in the graph
        return normalize(soft_dtw_autograd(cost.detach() / temp, end_ind), 1), cost
    w = soft_dtw((cost.detach() / temp.detach()).numpy(), end_ind.numpy())
    return normalize(torch.from_numpy(w), 1), cost


def averaging_loss(hp, sd, images_bf, traj_seq, match_dist_bf, pad_mask):
    """LossAveragingCriterion.loss (binding_loss.py:19-42), weights = 1: gaussian NLL of every (node, frame) pair weighted
    by the matching probability; value = sum over (node, frame), mean over batch."""
    d = batch_cdist(images_bf, traj_seq, reduction="sum")
    ls = sd["decoder.log_sigma"]
    n = float(np.prod(images_bf.shape[2:]))
    val = 0.5 * d * torch.exp(-ls) ** 2 + n * (ls + 0.5 * math.log(2 * math.pi))
    val = val * match_dist_bf * pad_mask[:, None]
    return val.sum((1, 2)).mean()


def soft_estimates(match_dist_bf, images_bf):
    """LossAveragingCriterion.get_soft_estimates (binding_loss.py:44-58) without the visualisation frame."""
    return torch.einsum("int,in...->it...", match_dist_bf, images_bf).detach()
