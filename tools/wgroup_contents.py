"""What the grouped weight-gradient launches of the c2 backward plan hold (name, rows, N, K, workgroups): python tools/wgroup_contents.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
hp = V.config(os.environ.get("CONFIG", "c2"))
model = GCPTreeModel(hp, device="cuda")
tr = GCPTrainStep(model)
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
tr.step({k: v.cuda() for k, v in inputs.items()})
torch.cuda.synchronize()
for g, items in tr.last_bplan.rec.get("_groups", {}).items():
    fl = sum(2.0 * r * n * k for _, r, n, k, _ in items)
    print(f"{g}: {len(items)} problems, {sum(i[4] for i in items)} workgroups, {fl / 1e9:.2f} GFLOP")
    for name, r, n, k, nb in sorted(items, key=lambda i: -i[1] * i[2] * i[3])[:6]:
        print(f"    {name:44s} R={r:6d} N={n:5d} K={k:5d} blocks={nb}")
