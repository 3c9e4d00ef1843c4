"""Golden vectors from the REFERENCE's hierarchical latent optimizer, produced by executing
/root/reference/gcp/planning/tree_optimizer.py (ImageHierarchicalTreeLatentOptimizer, :7-260) in the build container.

Run from the repo root:  python tests/golden/make_ref_planner_goldens.py  ->  tests/golden/ref_tree_optimizer.npz
`blox` is absent; the two container helpers the module imports (AttrDict, listdict2dictlist) are provided by an in-memory
stand-in below.  Stub simulator / cost functions: tests/golden/planner_stubs.py (shared with the parity test).
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from planner_stubs import StubCost, stub_rollouts, RES, D   # noqa: E402

REF = "/root/reference/gcp/planning/tree_optimizer.py"


def install_shim():
    class AttrDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    blox = types.ModuleType("blox")
    blox.AttrDict = AttrDict
    bt = types.ModuleType("blox.basic_types")
    bt.listdict2dictlist = lambda l: AttrDict({k: [d[k] for d in l] for k in l[0]})
    blox.basic_types = bt
    sys.modules["blox"], sys.modules["blox.basic_types"] = blox, bt


def main():
    install_shim()
    spec = importlib.util.spec_from_file_location("ref_tree_optimizer", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for case, (depth, rates, n_ll, ld, seed) in enumerate([(4, [3, 2], 2, 4, 0), (5, [4, 3], 3, 5, 1), (3, [], 4, 4, 2)]):
        np.random.seed(seed)
        cost = StubCost()
        opt = ref.ImageHierarchicalTreeLatentOptimizer(ld, list(rates), depth, cost, cost, n_ll)
        goal = np.random.rand(1, RES, RES, 3)               # raw env goal image (len(goal.shape) > 2 branch)
        out[f"c{case}_cfg"] = np.array([depth, n_ll, ld, seed] + list(rates))
        out[f"c{case}_goal"] = goal
        for it in range(len(rates) + 1):
            z = opt.sample()
            rollouts = stub_rollouts(z)
            best_rollout, best_cost = opt.optimize(rollouts, goal)
            out[f"c{case}_it{it}_z"] = z
            out[f"c{case}_it{it}_best_rollout"] = np.asarray(best_rollout)
            out[f"c{case}_it{it}_best_cost"] = np.asarray(best_cost, dtype=np.float64).reshape(-1)
            out[f"c{case}_it{it}_fully"] = np.array([bool(opt.fully_optimized)])
        out[f"c{case}_final_z"] = opt.sample()
    path = os.path.join(HERE, "ref_tree_optimizer.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    for k, v in out.items():
        print(k, v.shape)


if __name__ == "__main__":
    main()
