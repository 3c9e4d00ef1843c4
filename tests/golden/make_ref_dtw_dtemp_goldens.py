"""Golden vectors for the LEARNED matching temperature (hyperparameters.py:132, adaptive.py:19-21, :51), made by EXECUTING the
reference's soft_dtw under torch autograd here:  w = soft_dtw(cost.detach() / temp, end_ind)  exactly as adaptive.py:51 calls it,
then d (sum w * G) / d temp for a random G.  RESULT OF THAT EXECUTION: the reference's gradient is NaN in every case (the lattice always
holds cells no alignment reaches; their -inf accumulators give logsumexp / exp a NaN derivative, and NaN * 0 = NaN reaches `temp`) —
which is presumably why the one DTW conf (base_configs/gcp_adaptive.py:9) turns the option off.  The NaN is stored as evidence
(`c{i}_ref_autograd_dtemp`); what pins the derivative this repo computes is the reference's FORWARD executed at temp +- h
(`c{i}_w_plus`, `c{i}_w_minus`, h = 1/256): central differences of the reference's own outputs.

Run from the repo root:  python tests/golden/make_ref_dtw_dtemp_goldens.py
Writes tests/golden/ref_dtw_dtemp.npz (inputs + expected outputs only).  The blox stand-ins are those of make_ref_dtw_goldens.py.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_ref_dtw_goldens as base  # noqa: E402

H = 1.0 / 256
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_dtw_dtemp.npz")


def main():
    base.install_blox_shim()
    pd = base.load("gcp/prediction/models/adaptive_binding/probabilistic_dtw.py", "ref_probabilistic_dtw")
    rng = np.random.RandomState(7)
    out = {}
    cases = [(3, 7, 5, [4, 2, 3], 1.0), (2, 15, 12, [11, 6], 0.35), (2, 31, 20, [19, 9], 2.5), (1, 63, 40, [33], 0.8)]
    for i, (B, r, c, ends, t0) in enumerate(cases):
        cost = torch.tensor(rng.rand(B, r, c).astype(np.float32) * (2.0 if i % 2 else 0.7))
        e = torch.tensor(ends, dtype=torch.long)
        Gm = torch.tensor(rng.randn(B, r, c).astype(np.float32))
        temp = torch.nn.Parameter(t0 * torch.ones(1))
        w = pd.soft_dtw(cost.detach() / temp, e)
        (g,) = torch.autograd.grad((w * Gm).sum(), temp, retain_graph=True)
        out[f"c{i}_cost"], out[f"c{i}_end"], out[f"c{i}_G"], out[f"c{i}_temp"] = cost.numpy(), e.numpy(), Gm.numpy(), np.float32(t0)
        out[f"c{i}_w"], out[f"c{i}_ref_autograd_dtemp"] = w.detach().numpy(), g.numpy()
        with torch.no_grad():
            out[f"c{i}_w_plus"] = pd.soft_dtw(cost / (temp + H), e).numpy()
            out[f"c{i}_w_minus"] = pd.soft_dtw(cost / (temp - H), e).numpy()
    out["n"], out["h"] = np.array(len(cases)), np.float32(H)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v for k, v in out.items() if "dtemp" in k})


if __name__ == "__main__":
    main()
