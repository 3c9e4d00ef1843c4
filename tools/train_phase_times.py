"""Untraced timeline of the c2 backward plan: HIP timing events on every lane at each fork point (rocprofv3 slows the host
enough to distort the lanes): python tools/train_phase_times.py"""
import os, sys, ctypes as C
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
for _ in range(3):
    tr.step(dev_in)
torch.cuda.synchronize()
bplan, lib = tr.last_bplan, model.lib
ops, labels = [], []
last = "start"
for op in bplan.ops:
    if op[0] in ("@fork", "@join"):
        ops.append(("@mark", None, ("t", len(labels)), 0)); labels.append(f"{op[0]} after {last}")
    elif not op[0].startswith("@") and op[3] == 0:
        last = op[0]
    ops.append(op)
ops.append(("@mark", None, ("t", len(labels)), 0)); labels.append("end")
streams = tr._backward_streams()
evs = {}
def ev():
    e = C.c_void_p(); rt.check(lib.gcpx_event_create(C.byref(e)), "ev"); return e
def on_mark(tag, i):
    if tag != "t":
        return tr._on_mark(tag, i)
    es = evs.setdefault(i, [ev() for _ in streams])
    for e, s in zip(es, streams):
        rt.check(lib.gcpx_event_record(e, s), "rec")
e0 = ev()
for rep in range(3):
    out = model.forward(dev_in, "train", None)
    torch.cuda.synchronize()
    rt.check(lib.gcpx_event_record(e0, streams[0]), "rec")
    bplan.run(streams + [torch.cuda.current_stream().cuda_stream], ops=ops, on_mark=on_mark)
    torch.cuda.synchronize()
ms = C.c_float()
print("%-44s %9s %9s %9s   (ms since backward start; lane idle = its earlier work done)" % ("point", "lane0", "lane1", "lane2"))
for i, lab in enumerate(labels):
    t = []
    for e in evs[i]:
        rt.check(lib.gcpx_event_elapsed_ms(e0, e, C.byref(ms)), "el"); t.append(ms.value)
    print("%-44s %9.3f %9.3f %9.3f" % (lab[:44], *t))
