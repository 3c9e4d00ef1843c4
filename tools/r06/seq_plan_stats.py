"""What the gcp_sequential backward plan looks like to the host: ops per lane, control ops, segment graphs."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs
hp = V.config("c2")
tr = SequentialTrainStep(GCPSequentialModel(hp, device="cuda"))
inputs, noise, _ = make_inputs(hp, seed=3, variant="A")
dev = {k: v.cuda() for k, v in inputs.items()}
for _ in range(3):
    tr.step(dev)
torch.cuda.synchronize()
bp = tr.last_bplan
print("segment_graphs", tr.segment_graphs, "ranges", bp.rec.get("segment_ranges"), "caller_lane", bp.rec.get("caller_lane"))
for label, ops in (("plan", bp.ops), ("replayed", bp.rec.get("_segments") or bp.ops)):
    c = collections.Counter((o[0] if o[0].startswith("@") else f"launch lane {o[3]}") for o in ops)
    print(label, len(ops), dict(c))
import time
lanes = tr._backward_streams() + [torch.cuda.current_stream().cuda_stream]
for label, ops in (("plan", bp.ops), ("replayed", bp.rec.get("_segments") or bp.ops)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); bp.run(lanes, ops=ops); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(label, "host issue %.2f ms, until done %.2f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))

# ---- pieces of the backward on their own (eager, no events): where the 12 ms are ----
def timed(label, ops, reps=3, keep_events=False):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if keep_events:
            bp.run(lanes, ops=ops)
        else:
            for name, fn, args, lane in ops:
                fn(*args, lanes[0])
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-60s %5d ops  %7.2f ms  (host %.2f)" % (label, len(ops), best * 1e3, th * 1e3))
ops = bp.ops
is_chain = lambda n: ("_lstm" in n and n.startswith(("bw.dgrad", "bw.lstm"))) or n.startswith("bw.dgrad:gen") or "latent" in n
first = min(i for i, o in enumerate(ops) if is_chain(o[0]))
last = max(i for i, o in enumerate(ops) if is_chain(o[0]))
first_main = min(i for i, o in enumerate(ops) if o[0].startswith("bw.dgrad:gen"))
K = lambda sel: [o for o in sel if not o[0].startswith("@")]
timed("before the first chain launch (losses)", K(ops[:first]))
timed("decoder backward .. before the main-lane chain (lane 0 only)", [o for o in K(ops[first:first_main]) if o[3] == 0])
timed("prior chain (lane 1 launches that are chain ops)", [o for o in K(ops[first:last + 1]) if o[3] == 1 and is_chain(o[0])])
timed("other lane-1 work (decoder / prior weight gradients)", [o for o in K(ops[first:last + 1]) if o[3] == 1 and not is_chain(o[0])])
timed("main-lane chain (gen + inf lockstep), launches only", [o for o in K(ops[first_main:last + 1]) if o[3] == 0])
timed("main-lane chain with its events, nothing beside it", [o for o in ops[first_main:last + 1] if o[0].startswith("@") or o[3] == 0], keep_events=True)
timed("behind the chains", K(ops[last + 1:]))
timed("whole backward as planned", ops, keep_events=True)

# ---- what slows the main-lane chain: the chain phase with parts of lane 1 taken out ----
seg = ops[first:last + 1]
ctl = lambda o: o[0].startswith("@")
timed("chain phase: everything", seg, keep_events=True)
timed("chain phase: main lane + prior chain (no weight gradients)", [o for o in seg if ctl(o) or o[3] == 0 or is_chain(o[0])], keep_events=True)
timed("chain phase: main lane + lane-1 weight gradients (no prior chain)", [o for o in seg if ctl(o) or o[3] == 0 or not is_chain(o[0])], keep_events=True)
timed("chain phase: main lane only", [o for o in seg if ctl(o) or o[3] == 0], keep_events=True)
