"""Host enqueue time vs wall time per headline forward call (is the step host-bound?): python tools/host_time.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.synthetic import make_inputs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
hp = V.config("c2")
model = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda")
inputs, noise, _ = make_inputs(hp, seed=0, variant="B")
dinp = {}
for k in ("traj_seq", "I_0", "I_g", "end_ind"):
    buf = model.input_buffer(k, inputs[k].shape)
    buf.copy_(inputs[k])
    dinp[k] = buf
dnoise = noise.cuda()
import contextlib
ctx = torch.cuda.stream(model._stream) if os.environ.get("ON_MODEL_STREAM") else contextlib.nullcontext()
ctx.__enter__()
for _ in range(5):
    model(dinp, "train", noise=(None if os.environ.get("DRAW") else dnoise))
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for _ in range(steps):
    h0 = time.perf_counter()
    model(dinp, "train", noise=(None if os.environ.get("DRAW") else dnoise))
    host.append(time.perf_counter() - h0)
h_end = time.perf_counter()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
host.sort()
print(f"wall {wall / steps * 1e3:.3f} ms/step; host enqueue: median {host[len(host)//2]*1e3:.3f} ms, max {host[-1]*1e3:.3f}; host loop done after {(h_end - t0) / steps * 1e3:.3f} ms/step")
