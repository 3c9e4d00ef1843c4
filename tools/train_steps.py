"""N training steps at c2 (profiling target: rocprofv3 --kernel-trace --stats -- python3 tools/train_steps.py [steps])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd import runtime as _rt
if os.environ.get('GCPX_LIB'):          # same-box A/B against another build of the library
    _rt.LIB_PATH = os.environ['GCPX_LIB']
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hp = V.config(os.environ.get("CONFIG", "c2"))
model = GCPTreeModel(hp, device="cuda")
tr = GCPTrainStep(model)
model.use_graph = os.environ.get('NOGRAPH') is None
tr.backward_graph = os.environ.get('BGRAPH') is not None
tr.side_priority = int(os.environ.get('SIDEPRIO', '0'))
if os.environ.get('NOSIDE'):
    tr.side_lanes = False
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
# host issue time per step (async launches; no sync inside)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t1 = time.perf_counter(); tr.step(dev_in); ts.append(time.perf_counter() - t1); torch.cuda.synchronize()
print("host issue time per step: %.2f ms (min %.2f)" % (sum(ts) / len(ts) * 1e3, min(ts) * 1e3))
