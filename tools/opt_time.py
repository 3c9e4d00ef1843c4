"""Device time of the optimizer step alone (RAdam + re-pack + re-split of the split-f16 weights), c2: python tools/opt_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
hp = V.config("c2")
model = GCPTreeModel(hp, device="cuda")
tr = GCPTrainStep(model)
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
for _ in range(3):
    tr.step(dev_in)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(10):
        tr.optimizer_step()
    e1.record(); torch.cuda.synchronize()
    print("optimizer_step: %.3f ms (device, 10 back to back)" % (e0.elapsed_time(e1) / 10))
t0 = time.perf_counter()
for _ in range(10):
    tr.optimizer_step()
h = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
print("host issue %.3f ms" % (h * 1e3))
