"""gcp_sequential training step at the c2 shapes: forward alone, forward + backward, whole step (10 calls each, one sync)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs
hp = V.config("c2")
model = GCPSequentialModel(hp, device="cuda")
tr = SequentialTrainStep(model)
inputs, noise, _ = make_inputs(hp, seed=3, variant="A")
dev = {k: v.cuda() for k, v in inputs.items()}
for _ in range(3):
    tr.step(dev)
def timed(label, f, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); th = 0.0
    for _ in range(n):
        h0 = time.perf_counter(); f(); th += time.perf_counter() - h0
    torch.cuda.synchronize()
    print("%-28s %6.2f ms per call (host %5.2f)" % (label, (time.perf_counter() - t0) / n * 1e3, th / n * 1e3), flush=True)
for _ in range(2):
    timed("forward (train phase)", lambda: model.forward(dev, "train"))
    timed("forward + backward", lambda: tr.backward(dev))
    timed("optimizer step alone", lambda: tr.optimizer_step())
    timed("step", lambda: tr.step(dev))
