"""Golden vectors from the REFERENCE's hand-written planner costs and flat samplers, produced by executing
/root/reference/gcp/planning/cem/cost_fcn.py (CostFcn / EuclideanDistance / EuclideanPathLength / StepPathLength / L2ImageCost, :8-77)
and /root/reference/gcp/planning/cem/sampler.py (FlatCEMSampler / PDDMSampler, :33-71), plus the input conventions of
/root/reference/gcp/planning/cem/cem_simulator.py (GCPImageSimulator._env2planner / _postprocess_inputs, ActCondGCPImageSimulator, :72-104),
in the build container.

Run from the repo root:  python tests/golden/make_ref_costs_goldens.py  ->  tests/golden/ref_costs_samplers.npz   (arrays only)

`blox` is absent: its two container helpers the modules import (AttrDict, listdict2dictlist) come from the in-memory stand-in of
make_ref_planner_goldens.py.  cost_fcn.py also imports TestTimeCostModel (the learned cost's network; blox layers) at module level —
nothing executed here touches it, so the import is satisfied by an empty placeholder class; sampler.py imports the reference's own
tree_optimizer.py, which is loaded from its file.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_ref_planner_goldens import install_shim   # noqa: E402

REF = "/root/reference/gcp/planning"


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    install_shim()
    for pkg in ("gcp", "gcp.planning", "gcp.prediction", "gcp.prediction.models", "gcp.prediction.models.auxilliary_models"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    cm = types.ModuleType("gcp.prediction.models.auxilliary_models.cost_mdl")
    cm.TestTimeCostModel = type("TestTimeCostModel", (), {})
    sys.modules[cm.__name__] = cm
    load("gcp.planning.tree_optimizer", os.path.join(REF, "tree_optimizer.py"))
    C = load("ref_cost_fcn", os.path.join(REF, "cem", "cost_fcn.py"))
    S = load("ref_sampler", os.path.join(REF, "cem", "sampler.py"))
    out = {}
    rng = np.random.RandomState(7)

    # ---- state-space costs: candidates of different lengths [len_i, D], goal [D] ----
    D = 6
    lens = [5, 3, 1, 8]
    rolls = [rng.randn(l, D) for l in lens]
    goal = rng.randn(D)
    out["state_lens"], out["state_goal"] = np.array(lens), goal
    for i, r in enumerate(rolls):
        out[f"state_roll{i}"] = r
    cases = [("EuclideanDistance", True, 2.0), ("EuclideanDistance", False, 2.0), ("EuclideanDistance", True, 1.0),
             ("EuclideanPathLength", True, 1.0), ("EuclideanPathLength", True, 0.5),
             ("StepPathLength", False, 3.0), ("StepPathLength", True, 1.0)]
    out["state_cases"] = np.array([f"{n}|{int(d)}|{w}" for n, d, w in cases])
    for k, (name, dense, w) in enumerate(cases):
        out[f"state_cost{k}"] = np.asarray(getattr(C, name)(dense, w)([r.copy() for r in rolls], goal), dtype=np.float64)

    # ---- image cost: rollouts are (flat image ++ latent) rows, the goal a raw [1, R, R, 3] image in [0, 1] ----
    R, nz = 4, C.L2ImageCost.LATENT_SIZE
    img_lens = [4, 2, 6]
    img_rolls = [rng.rand(l, 3 * R * R + nz) * 2 - 1 for l in img_lens]
    goal_img = rng.rand(1, R, R, 3)
    out["img_lens"], out["img_goal"] = np.array(img_lens), goal_img
    for i, r in enumerate(img_rolls):
        out[f"img_roll{i}"] = r
    for k, (dense, w) in enumerate([(True, 1.0), (False, 1.0), (True, 4.0)]):
        out[f"img_cost{k}"] = np.asarray(C.L2ImageCost(dense, w)([r.copy() for r in img_rolls], goal_img), dtype=np.float64)
    out["img_cases"] = np.array([[1, 1.0], [0, 1.0], [1, 4.0]])

    # ---- samplers: np.random state -> populations, refits ----
    n, steps, ad = 6, 5, 3
    out["sampler_shape"] = np.array([n, steps, ad])
    for tag, cls in (("flat", S.FlatCEMSampler), ("pddm", S.PDDMSampler)):
        for clip in (np.inf, 0.8):
            s = cls(clip, steps, ad, 0.7)
            s.mean = rng.randn(steps, ad) * 0.3                    # a refit state, not the initial zeros
            s.std = 0.2 + rng.rand(steps, ad)
            ct = "inf" if np.isinf(clip) else "clip"
            out[f"{tag}_{ct}_mean"], out[f"{tag}_{ct}_std"] = s.mean.copy(), s.std.copy()
            np.random.seed(11)
            # the Gaussian numbers the call below consumes, in its own form: scale * standard normal (+ loc)
            out[f"{tag}_{ct}_unit_noise"] = np.random.standard_normal(size=(n, steps, ad))
            np.random.seed(11)
            out[f"{tag}_{ct}_samples"] = s.sample(n)
        data = rng.randn(n, steps, ad)
        scores = rng.rand(n) * 3
        s = cls(np.inf, steps, ad, 0.7)
        std0 = s.std.copy()
        s.fit(data, scores)
        out[f"{tag}_fit_data"], out[f"{tag}_fit_scores"] = data, scores
        out[f"{tag}_fit_mean"], out[f"{tag}_fit_std"], out[f"{tag}_fit_std_before"] = s.mean, s.std, std0
    # ---- simulator input conventions (cem_simulator.py:72-104) ----
    import torch
    SIM = load("ref_cem_simulator", os.path.join(REF, "cem", "cem_simulator.py"))
    AttrDict = sys.modules["blox"].AttrDict
    u8 = rng.randint(0, 256, size=(1, 4, 4, 3)).astype(np.uint8)
    unit = rng.rand(2, 4, 4, 3).astype(np.float32)                 # already in [0, 1]
    five = rng.randint(0, 256, size=(1, 2, 4, 4, 3)).astype(np.uint8)
    for tag, img in (("u8", u8), ("unit", unit), ("five", five)):
        out[f"env_{tag}_in"] = img
        out[f"env_{tag}_out"] = SIM.GCPImageSimulator._env2planner(torch.tensor(img.astype(np.float32))).numpy()
    acts = rng.randn(3, 5, 2).astype(np.float32)
    sim = SIM.ActCondGCPImageSimulator(None, True)
    inp = sim._postprocess_inputs(AttrDict(z=torch.tensor(acts), I_0=torch.tensor(u8.astype(np.float32)).repeat(3, 1, 1, 1),
                                           I_g=torch.tensor(u8.astype(np.float32)).repeat(3, 1, 1, 1)))
    assert "z" not in inp
    out["act_in"], out["act_actions"], out["act_pad_mask"], out["act_I_0"] = acts, inp.actions.numpy(), inp.pad_mask.numpy(), inp.I_0.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_costs_samplers.npz"), **out)
    print("wrote ref_costs_samplers.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
