"""Generate golden vectors by EXECUTING the reference's own functions in the build container.

Run from the repo root:  python tests/golden/make_ref_goldens.py
Writes tests/golden/ref_tree_utils.npz (inputs + expected outputs only — no reference source is copied).

The reference imports `blox` (un-vendored, empty submodule) for trivial tensor helpers; this script installs a
throw-away in-memory stand-in for exactly the helpers the executed functions touch (slice_tensor, reduce_dim, ...):
goldens inherit those (trivial) assumptions.  It also restores `np.int`/`np.float`, removed from NumPy >= 1.24
(tree_utils.py:225, utils.py:48).  Functions executed:
    gcp/prediction/utils/tree_utils.py:  depthfirst2layers (:222-232), depthfirst2breadthfirst (:217-219), interleave (:202-208)
    gcp/prediction/utils/utils.py:       get_pad_mask (:30-50)
BalancedBinding.comp_timestep is NOT executed: under torch 2.x its Long/Long division is true division, which is
not the torch-1.3 behaviour the reference pins; the midpoint rule is pinned by hand-derived KATs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_tree_utils.npz")


def install_blox_shim():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    blox = mod("blox")
    bt = mod("blox.torch"); ops = mod("blox.torch.ops"); porch = mod("blox.torch.porch")
    basic = mod("blox.basic_types"); tens = mod("blox.tensor"); tops = mod("blox.tensor.ops")
    blox.torch, blox.basic_types, blox.tensor = bt, basic, tens
    bt.ops, bt.porch, tens.ops = ops, porch, tops

    def slice_tensor(t, start, step, dim):
        idx = [slice(None)] * t.dim()
        idx[dim] = slice(start, None, step)
        return t[tuple(idx)]
    ops.slice_tensor = slice_tensor
    ops.reduce_dim = lambda t, dim: t
    porch.cat = torch.cat
    basic.map_dict = lambda fn, d: {k: fn(v) for k, v in d.items()}
    basic.listdict2dictlist = lambda l: {k: [d[k] for d in l] for k in l[0]}
    tops.batch_apply = lambda *a, **k: None
    tops.make_recursive_list = lambda fn: fn
    tops.rmap = lambda fn, x: fn(x)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    np.int, np.float = int, float          # removed aliases the reference still uses
    install_blox_shim()
    tu = load("gcp/prediction/utils/tree_utils.py", "ref_tree_utils")
    out = {}
    rng = np.random.RandomState(0)
    for n in (7, 15, 127):
        x = torch.tensor(rng.randint(0, 1000, size=(2, n, 3)))
        out[f"d2l_in_{n}"] = x.numpy()
        for i, l in enumerate(tu.depthfirst2layers(x, 1)):
            out[f"d2l_out_{n}_{i}"] = l.numpy()
        out[f"d2b_out_{n}"] = tu.depthfirst2breadthfirst(x, 1).numpy()
    a, b = torch.tensor(rng.randint(0, 100, size=(3, 4, 2))), torch.tensor(rng.randint(0, 100, size=(3, 4, 2)))
    out["il_a"], out["il_b"], out["il_out"] = a.numpy(), b.numpy(), tu.interleave(a, b).numpy()
    # get_pad_mask: numpy branch of utils.py:30-50 (the module imports cv2/dload at top: exec only that function)
    src = open(os.path.join(REF, "gcp/prediction/utils/utils.py")).read()
    start = src.index("def get_pad_mask")
    end = src.index("def datetime_str")
    ns = {"torch": torch, "np": np, "partial": __import__("functools").partial}
    exec(compile(src[start:end], "ref_utils_get_pad_mask", "exec"), ns)
    e = np.array([0, 5, 19, 7])
    out["pm_end"], out["pm_out"] = e, ns["get_pad_mask"](e, 20).astype(np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
