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

# ---- pieces of the backward on their own (eager, one stream, no events): where the 12 ms are ----
names = [o[0] for o in bp.ops]
def timed(label, ops, reps=3):
    st = lanes[0]
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for name, fn, args, lane in ops:
            fn(*args, st)
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-44s %5d launches  %7.2f ms  (host %.2f)" % (label, len(ops), best * 1e3, th * 1e3))
K = [o for o in bp.ops if not o[0].startswith("@")]
for net in ("prior_lstm", "gen_lstm", "inf_lstm"):
    timed(f"{net} chain alone", [o for o in K if (f":{net}" in o[0] and o[0].startswith(("bw.dgrad", "bw.lstm"))) and ".out" not in o[0].split(net)[0]])
first_chain = min(i for i, o in enumerate(bp.ops) if "prior_lstm" in o[0] or "gen_lstm" in o[0])
last_chain = max(i for i, o in enumerate(bp.ops) if o[0].startswith(("bw.dgrad:inf_lstm", "bw.lstm:inf_lstm", "bw.dgrad:gen_lstm")))
timed("before the chains (losses, decoder backward)", [o for o in bp.ops[:first_chain] if not o[0].startswith("@")])
timed("decoder weight gradients etc. (lane 1, not prior)", [o for o in bp.ops[first_chain:last_chain] if not o[0].startswith("@") and "_lstm" not in o[0] and "latent" not in o[0]])
timed("behind the chains (weight gradients, encoders)", [o for o in bp.ops[last_chain + 1:] if not o[0].startswith("@")])
print([o[0] for o in bp.ops[first_chain:last_chain] if not o[0].startswith("@") and "_lstm" not in o[0] and "latent" not in o[0]][:40])

# ---- the chain phase with its events, lanes taken out one at a time ----
def timed_plan(label, keep, reps=3):
    ops = [o for o in bp.ops[first_chain:last_chain + 1] if o[0].startswith("@") or keep(o[0])]
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bp.run(lanes, ops=ops)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-64s %5d ops %7.2f ms" % (label, len(ops), best * 1e3))
chain = lambda n, net: f":{net}" in n and n.startswith(("bw.dgrad", "bw.lstm"))
timed_plan("chain phase as planned", lambda n: True)
timed_plan("generator launches + all events", lambda n: chain(n, "gen_lstm"))
timed_plan("generator + inference (+ latent)", lambda n: chain(n, "gen_lstm") or chain(n, "inf_lstm") or "latent" in n)
timed_plan("generator + prior", lambda n: chain(n, "gen_lstm") or chain(n, "prior_lstm"))
timed_plan("three chains, no decoder weight gradients", lambda n: "_lstm" in n or "latent" in n)
ops_noev = [o for o in bp.ops[first_chain:last_chain + 1] if chain(o[0], "gen_lstm")]
timed("generator launches, no events (one stream)", ops_noev)
