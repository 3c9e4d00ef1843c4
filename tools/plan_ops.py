"""Backward plan of the c2 training step in issue order (name, lane, standalone device time): python tools/plan_ops.py [substr]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd import runtime as rt
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
hp = V.config("c2")
model = GCPTreeModel(hp, device="cuda")
tr = GCPTrainStep(model)
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
for _ in range(2):
    tr.step(dev_in)
torch.cuda.synchronize()
plan = tr.last_bplan
st = model._stream
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with torch.cuda.stream(st):
    for i, (nm, fn, args, lane) in enumerate(plan.ops):
        if nm.startswith("@"):
            print(f"{i:4d} {nm} {args[0]}")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rt.check(fn(*args, st.cuda_stream), nm)
        e0.record(st)
        for _ in range(3):
            rt.check(fn(*args, st.cuda_stream), nm)
        e1.record(st)
        st.synchronize()
        if pat in nm:
            print(f"{i:4d} lane{lane} {nm:44s} {1e3 * e0.elapsed_time(e1) / 3:8.1f} us")
