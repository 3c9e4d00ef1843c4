"""Soak: three c2 trainers on the same data and noise — two with the default step (early optimizer slices, merge chains on the caller's
stream, heads / gradient clear on side lanes), one with the optimizer behind the backward — must hold bit-identical parameters, moments
and packed weights after every step: the step is deterministic (fixed lanes, fixed summation orders), so a race between a slice and a
late reader of its weights, between lanes, or a stale pack shows as a difference.  (Other lane assignments change the ORDER in which
weight-gradient pieces are added and with it the last bits: those settings are compared through their tests' tolerances, not here.)
    python tools/soak_early_vs_late.py [steps=60]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hp = V.config("c2")
sd = V.init_params(hp, seed=3)
a = GCPTrainStep(GCPTreeModel(hp, params={k: v.clone() for k, v in sd.items()}, device="cuda"), lr=1e-3)
b = GCPTrainStep(GCPTreeModel(hp, params={k: v.clone() for k, v in sd.items()}, device="cuda"), lr=1e-3)
c = GCPTrainStep(GCPTreeModel(hp, params={k: v.clone() for k, v in sd.items()}, device="cuda"), lr=1e-3)
c.early_optimizer = False
for s in range(steps):
    inputs, noise, _ = make_inputs(hp, seed=100 + s % 7, variant="A")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    nz = noise.cuda()
    oa = a.step(dev_in, nz)
    for o_, name in ((b, "second default trainer"), (c, "late optimizer")):
        ob = o_.step(dev_in, nz)
        torch.cuda.synchronize()
        for x, y, what in [(a.m.theta, o_.m.theta, "theta"), (a.exp_avg, o_.exp_avg, "exp_avg"), (a.exp_avg_sq, o_.exp_avg_sq, "exp_avg_sq"),
                           (a.m._arena, o_.m._arena, "arena"), (oa.raw["losses"], ob.raw["losses"], "losses")]:
            if not torch.equal(x, y):
                print(f"step {s}, {name}: {what} differs, max |d| = {float((x - y).abs().max()):.3e}")
                sys.exit(1)
        ga = [v[0] for v in a.m._gsplit.values()]
        gb = [v[0] for v in o_.m._gsplit.values()]
        assert len(ga) == len(gb) and all(torch.equal(x, y) for x, y in zip(ga, gb)), f"step {s}, {name}: live split packs differ"
print(f"{steps} steps: parameters, moments, packed weights, live split packs and losses bit-identical; loss {float(oa.raw['losses'][5]):.4f}")
