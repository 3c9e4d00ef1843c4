"""Deterministic numpy stand-ins shared by the golden generator and the parity test of the hierarchical latent
optimizer: a stub simulator (samples -> rollouts) and stub cost functions with the call contracts of
LearnedCostEstimate (gcp/planning/cem/cost_fcn.py:79-101).  Test data only."""
import numpy as np

RES, D = 2, 3                      # image resolution (3*RES*RES state dims) and latent (cost input) dims


class StubCost:
    """pair branch: ndarray [n,D] x2 -> [n,1]; list branch: summed step cost per sequence -> [n]."""
    input_dim = D

    def __call__(self, a, b):
        if isinstance(a, np.ndarray):
            return np.sum((a - b) ** 2, axis=-1, keepdims=True) + 0.1 * np.abs(a[:, :1])
        costs = []
        for seq, goal in zip(a, b):
            full = np.concatenate((seq, goal.reshape(-1, seq.shape[-1])[:1]))
            costs.append(np.sum((full[1:] - full[:-1]) ** 2))
        return np.array(costs)


def stub_rollouts(z):
    """z [n, N, latent_dim] -> list of [N, 3*RES*RES + D] (image ++ latent), one frame per tree node."""
    n, N, ld = z.shape
    A = np.cos(np.arange(ld * 3 * RES * RES).reshape(ld, -1) * 0.37)
    Bm = np.sin(np.arange(ld * D).reshape(ld, D) * 0.91)
    t = np.linspace(0.0, 1.0, N)[None, :, None]
    img = np.tanh(z @ A) + t
    lat = z @ Bm + 2.0 * t
    return [np.concatenate((img[i], lat[i]), -1) for i in range(n)]


def flat_stub_len(sample, max_seq_len):
    """rollout length of ONE flat candidate [steps, ad]: a function of its content, so that chunked and whole-population rollouts agree"""
    return 3 + int(abs(float(sample[0, 0])) * 1000.0) % (max_seq_len - 2)


def flat_stub_rollout(samples, max_seq_len, dtype=np.float64):
    """samples [n, steps, ad] -> (predictions: list of [len_i, ad] paths = half the running sum of the candidate's steps, latents: the
    steps themselves), ragged lengths — the stub simulator of the flat CEM loop's fixtures (ref_cem_loop.npz)"""
    preds, lats = [], []
    for s in np.asarray(samples):
        n = flat_stub_len(s, max_seq_len)
        preds.append((0.5 * np.cumsum(s[:n], axis=0)).astype(dtype))
        lats.append(s[:n].astype(dtype))
    return preds, lats
