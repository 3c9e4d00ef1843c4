"""gcp_sequential forward, plain model against the trainer's model (save_for_backward): per-op-group device time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs
hp = V.config("c2")
inputs, noise, _ = make_inputs(hp, seed=3, variant="A")
dev = {k: v.cuda() for k, v in inputs.items()}
plain = GCPSequentialModel(hp, device="cuda")
trm = GCPSequentialModel(hp, device="cuda")
tr = SequentialTrainStep(trm)
tr.step(dev)
for label, m in (("plain", plain), ("trainer's", trm)):
    for _ in range(3):
        m.forward(dev, "train")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        m.forward(dev, "train")
    torch.cuda.synchronize()
    print(label, "forward %.2f ms" % ((time.perf_counter() - t0) * 100))
    res = m.profile_ops(dev, "train")
    groups, cnt = {}, {}
    for n, t in res:
        g = n.split(":")[0].rstrip("0123456789.")
        groups[g] = groups.get(g, 0) + t; cnt[g] = cnt.get(g, 0) + 1
    print("  ops", len(res), "sum %.0f us" % sum(t for _, t in res))
    for k, v in sorted(groups.items(), key=lambda x: -x[1])[:14]:
        print("   %-28s %4d launches %8.0f us" % (k, cnt[k], v))
