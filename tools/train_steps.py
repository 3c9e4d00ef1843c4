"""N training steps at c2 (profiling target: rocprofv3 --kernel-trace --stats -- python3 tools/train_steps.py [steps])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hp = V.config("c2")
model = GCPTreeModel(hp, device="cuda")
tr = GCPTrainStep(model)
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
for _ in range(3):
    tr.step(dev_in)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr.step(dev_in)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms/step")
