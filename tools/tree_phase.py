"""Wall time of the tree part of the backward plan (c2), all lanes / main lane only / side lanes only: tuning aid."""
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
for _ in range(2):
    tr.step(dev_in)
torch.cuda.synchronize()
plan = tr.last_bplan
names = [o[0] for o in plan.ops]
a = names.index("bw.dgrad:out6")
b = names.index("bw.dgrad:seq.head")
ops = plan.ops[a:b]
streams = tr._backward_streams()
def timed(sub, tag):
    sub = list(sub) + [plan.ops[-1]]          # final join of the side lanes
    for _ in range(2):
        plan.run(streams, sub)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        plan.run(streams, sub)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{tag:28s} {1e3 * (time.perf_counter() - t0) / 5:7.2f} ms wall   host issue {1e3 * th / 5:6.2f} ms   {len(sub)} ops")
timed(ops, "tree phase, all lanes")
timed([o for o in ops if o[0].startswith("@") or o[3] == 0], "main lane only")
timed([o for o in ops if o[0].startswith("@") or o[3] != 0], "side lanes only")
timed([o for o in ops if not o[0].startswith("@") and o[3] == 0], "main lane, no fork events")
for lvl in (6, 3, 0):
    s = names.index(f"bw.dgrad:out{lvl}")
    e = names.index(f"bw.dgrad:out{lvl - 1}") if lvl else names.index("bw.dgrad:lstm_init.out")
    timed(plan.ops[s:e], f"level {lvl} all lanes")
    timed([o for o in plan.ops[s:e] if not o[0].startswith("@") and o[3] == 0], f"level {lvl} main only")
