"""Generate golden vectors for the adaptive-binding path by EXECUTING the reference's own DTW functions here.

Run from the repo root:  python tests/golden/make_ref_dtw_goldens.py
Writes tests/golden/ref_dtw.npz (inputs + expected outputs only — no reference source is copied).

Functions executed (paths relative to /root/reference):
    gcp/prediction/models/adaptive_binding/probabilistic_dtw.py:  fast_gak (:11-73), soft_dtw (:82-122)
    gcp/evaluation/dtw_utils.py:                                   basic_dtw (:77-95) incl. _traceback (:201-218)
`probabilistic_dtw.py` imports three helpers from the un-vendored `blox` submodule; this script installs a throw-away
in-memory stand-in for exactly those (the goldens inherit these — trivial — assumptions):
    blox.torch.ops.batchwise_index(t, inds)        -> t[arange(B), inds]          (index dim 1 per batch element)
    blox.torch.ops.batchwise_assign(t, inds, val)  -> t[arange(B), inds] = val
    blox.tensor.ndim.stack / .flip                 -> torch.stack / torch.flip
`dtw_utils.py` imports as-is (numpy / scipy / torch only; its optional Cython module is absent and guarded by try/except).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_dtw.npz")


def install_blox_shim():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    blox = mod("blox")
    tens, bt, ops, utils = mod("blox.tensor"), mod("blox.torch"), mod("blox.torch.ops"), mod("blox.utils")
    blox.tensor, blox.torch, blox.utils, bt.ops = tens, bt, utils, ops
    ndim = types.SimpleNamespace(stack=torch.stack, flip=torch.flip)
    tens.ndim = ndim

    def batchwise_index(t, inds):
        return t[torch.arange(t.shape[0]), inds]

    def batchwise_assign(t, inds, val):
        t[torch.arange(t.shape[0]), inds] = val
    ops.batchwise_index, ops.batchwise_assign = batchwise_index, batchwise_assign
    utils.timing = lambda *a, **k: (lambda f: f)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    install_blox_shim()
    pd = load("gcp/prediction/models/adaptive_binding/probabilistic_dtw.py", "ref_probabilistic_dtw")
    du = load("gcp/evaluation/dtw_utils.py", "ref_dtw_utils")
    out = {}
    rng = np.random.RandomState(0)
    # soft_dtw: cost [B, r = nodes, c = frames] (r >= c), end_inds [B]
    cases = [(3, 7, 5, [4, 2, 3]), (2, 15, 12, [11, 6]), (2, 31, 20, [19, 9]), (1, 63, 40, [33])]
    for i, (B, r, c, ends) in enumerate(cases):
        cost = torch.tensor(rng.rand(B, r, c).astype(np.float32) * (2.0 if i % 2 else 0.7))
        e = torch.tensor(ends, dtype=torch.long)
        w = pd.soft_dtw(cost, e)
        out[f"sd{i}_cost"], out[f"sd{i}_end"], out[f"sd{i}_w"] = cost.numpy(), e.numpy(), w.numpy()
        # the forward accumulator alone (fast_gak on -cost, begin index 0), float64
        acc = pd.fast_gak((-cost).double(), transition="nohor", begin_inds=torch.zeros(B, dtype=torch.long))
        out[f"sd{i}_fwd"] = acc.numpy()
    out["sd_n"] = np.array(len(cases))
    # hard DTW of the evaluation harness (dtw_utils.py): accumulated cost, normalised distance, path
    for i, (r, c) in enumerate([(7, 5), (12, 12), (20, 31)]):
        Cm = rng.rand(r, c)
        d, D, path = du.basic_dtw(Cm)
        out[f"bd{i}_C"], out[f"bd{i}_d"], out[f"bd{i}_D"] = Cm, np.array(d), D
        out[f"bd{i}_p0"], out[f"bd{i}_p1"] = np.asarray(path[0]), np.asarray(path[1])
    out["bd_n"] = np.array(3)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("_w") or k.endswith("_D")})


if __name__ == "__main__":
    main()
