"""N forwards (+ losses) of the adaptive model at c5 (profiling target: rocprofv3 --kernel-trace --stats -- python3 tools/adaptive_fwd.py [n])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from helpers import make_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hp = V.config("c5")
model = GCPTreeModel(hp, device="cuda")
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
noise = noise.cuda()
for _ in range(3):
    model(dev_in, "train", noise=noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    model(dev_in, "train", noise=noise)
torch.cuda.synchronize()
print(f"adaptive forward (c5, B={hp.batch_size}): {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
