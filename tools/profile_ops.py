"""Per-op device times of the c2 forward (each op launched back-to-back `repeats` times between two events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from helpers import make_inputs
hp = V.config("c2")
model = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda")
inputs, noise, _ = make_inputs(hp, seed=100, variant="A")
dinp = {k: v.cuda() for k, v in inputs.items()}
res = model.profile_ops(dinp, "train", noise=noise.cuda())
tot = sum(t for _, t in res)
groups = {}
for n, t in res:
    g = n.split(":")[0].rstrip("0123456789.")
    groups[g] = groups.get(g, 0) + t
for n, t in res:
    print(f"{n:28s} {t:9.1f} us")
print("---- by group")
for g, t in sorted(groups.items(), key=lambda x: -x[1]):
    print(f"{g:28s} {t:9.1f} us  {100*t/tot:5.1f}%")
print("total", tot, "us over", len(res), "ops")
