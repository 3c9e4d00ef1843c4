"""Wall time of the tree part of the c2 FORWARD plan (length predictor .. last level's out GEMM): whole phase and per level, replayed
as a captured hipGraph (what the model does) and eagerly; plus the per-op standalone device times.  Tuning aid for the level chain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd import runtime as rt
from video_gcp_amd.model import GCPTreeModel
from helpers import make_inputs
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
hp = V.config(cfg)
model = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda")
inputs, noise, _ = make_inputs(hp, seed=100, variant="A")
dinp = {k: v.cuda() for k, v in inputs.items() if k in ("traj_seq", "I_0", "I_g", "end_ind", "start_ind")}
for _ in range(3):
    model(dinp, "train", noise=noise.cuda())
torch.cuda.synchronize()
plan = [v[1] for v in model._plans.values()][-1]
names = [o[0] for o in plan.ops]
L = hp.hierarchy_levels
def lvl_start(l):
    hits = [i for i, nm in enumerate(names) if nm in (f"prior{l}", f"posterior{l}", f"merge{l}") or nm.startswith(f"prior+posterior{l}")
            or nm.endswith(f"+merge{l}")]
    if not hits:
        raise KeyError(l)
    return min(hits)


first = lvl_start(0)
# start at the fork that precedes the first tree op
a = first
while a > 0 and names[a - 1].startswith("@"):
    a -= 1
b = names.index(f"out{L - 1}") + 1
st = model._stream
stream = st.cuda_stream


def timed(ops, tag, reps=20):
    with torch.cuda.stream(st):
        plan.run(model._streams, ops)
        g = model._capture(plan, ops, stream)
        for _ in range(3):
            rt.check(model.lib.gcpx_graph_launch(g, stream), "launch")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            rt.check(model.lib.gcpx_graph_launch(g, stream), "launch")
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.run(model._streams, ops)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / reps
    n = sum(1 for o in ops if not o[0].startswith("@"))
    print(f"{tag:34s} graph {1e6 * tg:8.1f} us   eager {1e6 * te:8.1f} us   {n} launches")


timed(plan.ops[a:b], "tree phase (all levels)")
for l in range(L):
    s = lvl_start(l)
    while s > 0 and names[s - 1].startswith("@"):
        s -= 1
    e = [i for i, nm in enumerate(names) if nm == f"out{l}" or nm.startswith(f"out{l}+")][0] + 1
    # (round 4: the wide levels' merge is forked on lane 1 inside the previous level's slice and joined in this one — a slice with an
    # unmatched fork / join cannot be captured on its own; its one-lane replay below still can)
    if sum(o[0] == "@fork" for o in plan.ops[s:e]) == sum(o[0] == "@join" for o in plan.ops[s:e]) and not any(o[0] == "@wait" for o in plan.ops[s:e]):
        timed(plan.ops[s:e], f"level {l}")
    else:
        print(f"level {l}: shares a side lane with its neighbour (merge of the wide level): whole-phase and one-lane figures only")
    timed([o for o in plan.ops[s:e] if not o[0].startswith("@")], f"level {l}, one lane, no events")
print("---- standalone per-op device time (5 back-to-back launches each)")
with torch.cuda.stream(st):
    for nm, fn, args, lane in plan.ops[a:b]:
        if nm.startswith("@"):
            print(f"      {nm} {args[0]}")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rt.check(fn(*args, stream), nm)
        e0.record(st)
        for _ in range(5):
            rt.check(fn(*args, stream), nm)
        e1.record(st)
        st.synchronize()
        print(f"lane{lane} {nm:26s} {1e3 * e0.elapsed_time(e1) / 5:8.1f} us")
