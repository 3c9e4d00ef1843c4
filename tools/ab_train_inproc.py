"""Same-process A/B of the c2 training step under two values of one trainer attribute or environment setting (box-to-box and
run-to-run spreads of 3-5 % hide effects of 1-2 %: four trainers live in one process, built A B B A, and take turns).
    python tools/ab_train_inproc.py NAME valueA valueB [rounds] [steps]
NAME = an attribute of GCPTrainStep set after construction (early_blocks, early_optimizer, ...; values are parsed as int) or, with a
leading '$', an environment variable read while the model / trainer is built ('-' = unset)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs

var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 8
hp = V.config("c2")
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
vals4 = (va, vb, vb, va)
trainers = []
for val in vals4:
    if var.startswith("$"):
        if val == "-":
            os.environ.pop(var[1:], None)
        else:
            os.environ[var[1:]] = val
    tr = GCPTrainStep(GCPTreeModel(hp, device="cuda"))
    if not var.startswith("$"):
        assert hasattr(tr, var), var
        setattr(tr, var, int(val))
    for _ in range(3):
        tr.step(dev_in)
    trainers.append(tr)
torch.cuda.synchronize()
res = [[], [], [], []]
for r in range(rounds):
    for i in ((0, 1, 2, 3) if r % 2 == 0 else (3, 2, 1, 0)):
        tr = trainers[i]
        tr.step(dev_in)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(dev_in)
        torch.cuda.synchronize()
        res[i].append(1e3 * (time.perf_counter() - t0) / steps)
med = lambda t: sorted(t)[len(t) // 2]
for val, idx in ((va, (0, 3)), (vb, (1, 2))):
    both = res[idx[0]] + res[idx[1]]
    print(f"{var}={val:8s} mean of the two trainers' medians {(med(res[idx[0]]) + med(res[idx[1]])) / 2:.3f} ms  "
          f"(built {idx[0] + 1}.: {med(res[idx[0]]):.3f}, built {idx[1] + 1}.: {med(res[idx[1]]):.3f})  min {min(both):.3f}  max {max(both):.3f}")
