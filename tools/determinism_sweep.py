"""Run-to-run determinism of the full-size paths: every tensor the headline forward returns, the adaptive (c5) and flat-VRNN training
steps' gradients — bit-identical across three runs on the same inputs (no float atomics anywhere: a difference is a race or a hazard).
    python tools/determinism_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs

def tensors(o, pre=""):
    out = {}
    for k, v in (o.items() if hasattr(o, "items") else []):
        if torch.is_tensor(v):
            out[pre + k] = v
        elif isinstance(v, dict):
            out.update(tensors(v, pre + k + "."))
    return out

def same(a, b):
    return a.shape == b.shape and bool(((a == b) | (torch.isnan(a) & torch.isnan(b)) if a.is_floating_point() else (a == b)).all())

bad = 0
hp = V.config("c2")
m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda"); m.train(True)
inputs, noise, _ = make_inputs(hp, seed=1, variant="A")
d = {k: v.cuda() for k, v in inputs.items()}
runs = []
for _ in range(3):
    o = m(d, "train", noise=noise.cuda()); torch.cuda.synchronize()
    runs.append({k: v.clone() for k, v in tensors(o.raw).items()})
diff = [k for k in runs[0] if not (same(runs[0][k], runs[1][k]) and same(runs[0][k], runs[2][k]))]
print("c2 headline forward:", len(runs[0]), "tensors, differing:", diff); bad += len(diff)
m.eval()
runs = []
for _ in range(3):
    o = m(d, "inference", noise=noise.cuda()); torch.cuda.synchronize()
    runs.append({k: v.clone() for k, v in tensors(o.raw).items()})
diff = [k for k in runs[0] if not (same(runs[0][k], runs[1][k]) and same(runs[0][k], runs[2][k]))]
print("c2 eval forward:", len(runs[0]), "tensors, differing:", diff); bad += len(diff)
del m
for name in ("c5",):
    hp = V.config(name)
    tr = GCPTrainStep(GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda"))
    inputs, noise, _ = make_inputs(hp, seed=2, variant="A")
    d = {k: v.cuda() for k, v in inputs.items()}
    gs = []
    for _ in range(3):
        tr.backward(d, noise.cuda()); torch.cuda.synchronize(); gs.append(tr.grad.clone())
    ok = torch.equal(gs[0], gs[1]) and torch.equal(gs[0], gs[2])
    print(name, "training gradient deterministic:", ok); bad += not ok
    del tr
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
hp = V.config("c2")
tr = SequentialTrainStep(GCPSequentialModel(hp, device="cuda"))
inputs, noise, _ = make_inputs(hp, seed=3, variant="A")
d = {k: v.cuda() for k, v in inputs.items()}
nz = noise[:, :hp.max_seq_len - 1].contiguous().cuda()
gs = []
for _ in range(3):
    tr.backward(d, nz); torch.cuda.synchronize(); gs.append(tr.grad.clone())
ok = torch.equal(gs[0], gs[1]) and torch.equal(gs[0], gs[2])
print("gcp_sequential training gradient deterministic:", ok); bad += not ok
print("OK" if not bad else f"{bad} nondeterministic results")
