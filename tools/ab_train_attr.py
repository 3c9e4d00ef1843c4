"""A/B of the c2 training step under values of ONE trainer attribute, on ONE trainer in ONE process: the attribute is flipped, the
backward plan rebuilt (same model, same streams, same hardware queues — separate processes differ by 3-5 %, separate trainers in a
process by the queues their streams land on), 2 warm-up + STEPS timed steps per turn, ROUNDS turns per value in alternating order.
    python tools/ab_train_attr.py ATTR value value [value ...] [rounds=6] [steps=8]
ATTR: merge_on_caller_lane, heads_on_side_lane, batch_dh, group_mlp_bwd, early_optimizer, early_blocks, dec_side_level, ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs

args = [a for a in sys.argv[1:] if "=" not in a]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
attr, vals = args[0], [int(v) for v in args[1:]]
rounds, steps = int(opts.get("rounds", 6)), int(opts.get("steps", 8))
hp = V.config(opts.get("cfg", "c2"), **({"batch_size": int(opts["batch"])} if "batch" in opts else {}))
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
tr = GCPTrainStep(GCPTreeModel(hp, device="cuda"))
on_model = attr.startswith("m.")              # "m.NAME": an attribute of the model (forward plans are rebuilt too)
obj, attr_ = (tr.m, attr[2:]) if on_model else (tr, attr)
assert hasattr(obj, attr_), attr
res = {v: [] for v in vals}
for r in range(rounds):
    for v in (vals if r % 2 == 0 else vals[::-1]):
        setattr(obj, attr_, v)
        if on_model:
            tr.m._clear_plans()
        tr._bplans.clear()
        for _ in range(2):
            tr.step(dev_in)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(dev_in)
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t0) / steps)
for v in vals:
    t = sorted(res[v])
    print(f"{attr}={v:<6d} median {t[len(t) // 2]:.3f} ms   min {t[0]:.3f}   max {t[-1]:.3f}   ({' '.join('%.2f' % x for x in res[v])})")
