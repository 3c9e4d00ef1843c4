"""gcp_sequential training step at the c2 shapes: eager backward plan (default) against the backward replayed as a hipGraph
(BGRAPH=1): python tools/bench_sequential_train.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hp = V.config("c2")
model = GCPSequentialModel(hp, device="cuda")
tr = SequentialTrainStep(model)
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
for mode in ([False, True] if os.environ.get("BGRAPH") else [False]):
    tr.backward_graph = mode
    for _ in range(3):
        tr.step(dev_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(dev_in)
    torch.cuda.synchronize()
    print(f"backward_graph={mode}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step")
    ts = []
    for _ in range(3):
        t1 = time.perf_counter(); tr.step(dev_in); ts.append(time.perf_counter() - t1); torch.cuda.synchronize()
    print("   host issue time per step: %.2f ms" % (min(ts) * 1e3))
