"""What the grouped weight-gradient launches of the adaptive (c5) backward hold, with each group's device time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd import runtime as rt
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
hp = V.config(sys.argv[1] if len(sys.argv) > 1 else "c5")
model = GCPTreeModel(hp, device="cuda")
tr = GCPTrainStep(model)
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev = {k: v.cuda() for k, v in inputs.items()}
for _ in range(2):
    tr.step(dev)
torch.cuda.synchronize()
plan = tr.last_bplan
st = model._stream
for nm, fn, args, lane in plan.ops:
    if not nm.startswith("bw.wgroup"):
        continue
    with torch.cuda.stream(st):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rt.check(fn(*args, st.cuda_stream), nm)
        e0.record(st)
        for _ in range(3):
            rt.check(fn(*args, st.cuda_stream), nm)
        e1.record(st); st.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 3
    items = plan.rec["_groups"][nm.split(":", 1)[1]]
    print(f"{nm:28s} lane {lane} {us:8.1f} us  blocks {sum(i[4] for i in items)}")
    if us > 100:
        for it in items[:8]:
            print("      %-44s R=%-7d N=%-5d K=%-5d blocks=%d" % it)
