"""Upper bound of what removing the bn_finalize launches from the forward's chains could give: the c2 headline forward timed as it is,
then with every `bn_finalize:*` op dropped from the recorded plan (scale / shift keep the previous forward's values: same work in every
other kernel), in turns.   python tools/bn_finalize_bound.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from helpers import make_inputs
hp = V.config("c2")
m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda")
m.force_replay("eager")
inputs, noise, _ = make_inputs(hp, seed=1, variant="A")
d = {k: v.cuda() for k, v in inputs.items()}
for _ in range(3):
    m(d, "train")
torch.cuda.synchronize()
plan = [v[1] for v in m._plans.values()][-1]
full = list(plan.ops)
nofin = [op for op in full if not op[0].startswith("bn_finalize")]
print(f"{len(full)} ops, {len(full) - len(nofin)} bn_finalize launches")
def timed(ops, n=30):
    plan.ops = ops
    m(d, "train"); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m(d, "train")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for r in range(3):
    print(f"round {r}: as recorded {timed(full):.3f} ms   without bn_finalize {timed(nofin):.3f} ms")
