"""Golden vectors from the REFERENCE's CEM loop itself, produced by executing /root/reference/gcp/planning/cem/cem_planner.py
(CEMPlanner.__call__ / _rollout / _get_best_rollouts, :55-135; HierarchicalCEMPlanner, :166-218) in the build container.

Run from the repo root:  python tests/golden/make_ref_cem_loop_goldens.py  ->  tests/golden/ref_cem_loop.npz   (arrays only)

`blox` is absent: AttrDict / listdict2dictlist come from the stand-in of make_ref_planner_goldens.py, ParamDict (an attribute
dict whose `overwrite` refuses unknown keys) is added here.  The simulator is a stub (tests/golden/planner_stubs.py: a deterministic
function of the candidate) and so is the learned cost of the hierarchical planner; the flat planner scores with the reference's own
EuclideanPathLength.  The Gaussian numbers each `sampler.sample` consumed are recorded by wrapping np.random.normal as
loc + scale * np.random.standard_normal (numpy's legacy `normal` is exactly that; asserted below against an unwrapped run).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_ref_planner_goldens import install_shim   # noqa: E402
from make_ref_costs_goldens import load              # noqa: E402
from planner_stubs import StubCost, stub_rollouts, flat_stub_rollout, RES, D   # noqa: E402

REF = "/root/reference/gcp/planning"


def install_all():
    install_shim()
    AttrDict = sys.modules["blox"].AttrDict

    class ParamDict(AttrDict):
        def overwrite(self, new):
            for k, v in new.items():
                if k not in self:
                    raise ValueError(f"unknown hyper-parameter {k}")
                self[k] = v
            return self
    bu = types.ModuleType("blox.utils")
    bu.ParamDict = ParamDict
    sys.modules["blox.utils"] = bu
    sys.modules["blox"].utils = bu
    for pkg in ("gcp", "gcp.planning", "gcp.planning.cem", "gcp.prediction", "gcp.prediction.models",
                "gcp.prediction.models.auxilliary_models"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    cm = types.ModuleType("gcp.prediction.models.auxilliary_models.cost_mdl")
    cm.TestTimeCostModel = type("TestTimeCostModel", (), {})
    sys.modules[cm.__name__] = cm
    load("gcp.planning.tree_optimizer", os.path.join(REF, "tree_optimizer.py"))
    C = load("gcp.planning.cem.cost_fcn", os.path.join(REF, "cem", "cost_fcn.py"))
    S = load("gcp.planning.cem.sampler", os.path.join(REF, "cem", "sampler.py"))
    P = load("gcp.planning.cem.cem_planner", os.path.join(REF, "cem", "cem_planner.py"))
    return AttrDict, ParamDict, C, S, P


class NoiseTap:
    """np.random.normal -> loc + scale * standard_normal(size), the unit numbers kept"""

    def __init__(self):
        self.eps, self._orig = [], np.random.normal

    def __enter__(self):
        def normal(loc=0.0, scale=1.0, size=None):
            e = np.random.standard_normal(size)
            self.eps.append(e)
            return loc + scale * e
        np.random.normal = normal
        return self

    def __exit__(self, *a):
        np.random.normal = self._orig


def main():
    AttrDict, ParamDict, C, S, P = install_all()
    out = {}

    class FlatStubSim:
        def __init__(self, dtype):
            self.dtype, self.calls = dtype, []

        def rollout(self, state, goal, samples, max_seq_len):
            self.calls.append(int(samples.shape[0]))
            preds, lat = flat_stub_rollout(samples, max_seq_len, self.dtype)
            return AttrDict(predictions=preds, states=[p.copy() for p in preds], actions=[p[1:] - p[:-1] for p in preds], latents=lat)

    # ---- flat CEM: 3 iterations; (batch, elite_frac, max_rollout_bs, rollout dtype, clip) ----
    flat_cases = [(40, 0.2, 100, np.float64, np.inf, 0), (64, 0.1, 100, np.float32, 0.5, 1), (48, 0.25, 16, np.float64, np.inf, 2)]
    out["flat_cases"] = np.array([[b, e, m, 8 if dt is np.float64 else 4, c, s] for b, e, m, dt, c, s in flat_cases])
    steps, ad = 6, 3
    out["flat_shape"] = np.array([steps, ad])
    for ci, (batch, efrac, mrb, dt, clip, seed) in enumerate(flat_cases):
        goal = np.linspace(-0.5, 0.5, ad).astype(dt)
        out[f"flat{ci}_goal"] = goal
        hp = dict(horizon=steps, action_dim=ad, n_iters=3, batch_size=batch, max_rollout_bs=mrb, elite_frac=efrac,
                  cost_fcn=C.EuclideanPathLength, dense_cost=True, final_step_cost_weight=2.0, sampler=S.FlatCEMSampler,
                  sampler_clip_val=clip, initial_std=0.6, max_seq_len=steps, use_inferred_actions=True)
        # unwrapped run: what the reference returns on its own random stream
        np.random.seed(seed)
        plain = P.CEMPlanner(dict(hp), FlatStubSim(dt))
        want = plain(None, goal)
        # wrapped run: the same stream, unit numbers kept, selections logged
        np.random.seed(seed)
        sim = FlatStubSim(dt)
        planner = P.CEMPlanner(dict(hp), sim)
        picks = []
        orig = planner._get_best_rollouts

        def tapped(rollouts, goal_state, samples, _o=orig, _p=picks, _pl=planner):
            r = _o(rollouts, goal_state, samples)
            _p.append((np.asarray(_pl._cost_fcn(rollouts.predictions, goal_state), dtype=np.float64), np.asarray(r[4]), np.asarray(r[2], dtype=np.float64),
                       np.asarray(r[3])))
            return r
        planner._get_best_rollouts = tapped
        with NoiseTap() as tap:
            got = planner(None, goal)
        for a, b in zip(want, got):
            assert np.array_equal(np.asarray(a), np.asarray(b)), "np.random.normal is not loc + scale * standard_normal here"
        for it in range(3):
            scores, idx, best_scores, best_samples = picks[it]
            assert len(np.unique(scores)) == len(scores), "tied scores: argsort order would be implementation-defined"
            out[f"flat{ci}_it{it}_eps"] = tap.eps[it]
            out[f"flat{ci}_it{it}_scores"] = scores                      # (only the candidates the chunked rollout covered, :114-121)
            out[f"flat{ci}_it{it}_elite_idx"] = idx
            out[f"flat{ci}_it{it}_elite_scores"] = best_scores
            out[f"flat{ci}_it{it}_elite_samples"] = best_samples
            d = planner._logs[0][it].dists
            out[f"flat{ci}_it{it}_mean"], out[f"flat{ci}_it{it}_std"] = np.asarray(d.mean), np.asarray(d.std)
        out[f"flat{ci}_rollout_calls"] = np.array(sim.calls)
        out[f"flat{ci}_plan_pred"], out[f"flat{ci}_plan_actions"], out[f"flat{ci}_plan_latents"] = (np.asarray(got[0]), np.asarray(got[1]), np.asarray(got[2]))
        out[f"flat{ci}_plan_score"] = np.asarray(got[3], dtype=np.float64).reshape(1)

    # ---- hierarchical CEM: the whole __call__ (cem_planner.py:166-218 over :55-96) on the stub simulator / stub learned cost ----
    class TreeStubSim:
        def rollout(self, state, goal, samples, max_seq_len):
            r = stub_rollouts(np.asarray(samples))
            n_img = 3 * RES * RES
            return AttrDict(predictions=r, states=[x[:, :2].copy() for x in r], actions=[x[1:, :2] - x[:-1, :2] for x in r],
                            latents=[x[:, n_img:].copy() for x in r])

    hier_cases = [(4, [3, 2], 2, 4, 0), (5, [4, 3, 2], 3, 5, 3)]
    out["hier_cases"] = np.array([[d, n, ld, s, len(r)] + r + [0] * (3 - len(r)) for d, r, n, ld, s in hier_cases])
    for ci, (depth, rates, n_ll, ld, seed) in enumerate(hier_cases):
        np.random.seed(seed)
        goal = np.random.rand(1, RES, RES, 3)
        hp = dict(action_dim=ld, n_iters=len(rates) + 1, batch_size=10, cost_fcn=lambda cfg: StubCost(), cost_config={},
                  sampler=S.ImageHierarchicalTreeCEMSampler, n_level_hierarchy=depth, sampling_rates_per_layer=list(rates),
                  n_ll_samples=n_ll, max_seq_len=2 ** depth - 1, sampler_clip_val=np.inf, initial_std=1.0)
        planner = P.HierarchicalCEMPlanner(dict(hp), TreeStubSim())
        np.random.seed(seed + 100)
        pred, actions, latents, score = planner(None, goal)
        out[f"hier{ci}_goal"] = goal
        out[f"hier{ci}_plan_pred"], out[f"hier{ci}_plan_actions"], out[f"hier{ci}_plan_latents"] = np.asarray(pred), np.asarray(actions), np.asarray(latents)
        out[f"hier{ci}_plan_score"] = np.asarray(score, dtype=np.float64).reshape(-1)[:1]
        for it, log in enumerate(planner._logs[0][:-1]):
            out[f"hier{ci}_it{it}_elite_rollout"] = np.asarray(log.elite_rollouts[0])
            out[f"hier{ci}_it{it}_elite_score"] = np.asarray(log.elite_scores, dtype=np.float64).reshape(-1)
        out[f"hier{ci}_fully"] = np.array([bool(planner._sampler.fully_optimized)])
    path = os.path.join(HERE, "ref_cem_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
