"""Which parameter gradients move when the encoder runs split-f16? (debugging aid for tests/test_gpu_training.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
from oracle import gcp_model_oracle as O
hp = V.config("c1", batch_size=3)
sd = V.init_params(hp, seed=1, randomize_affine=True)
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 3
inputs, noise, _ = make_inputs(hp, seed=SEED, variant="A")
gref, res, total, _ = O.gradients(sd, hp, inputs, noise)
model = GCPTreeModel(hp, params=sd, device="cuda")
model.use_graph = False
tr = GCPTrainStep(model, lr=1e-3)
out = tr.backward({k: v.cuda() for k, v in inputs.items()}, noise.cuda())
torch.cuda.synchronize()
got = tr.named_grads()
if 0: print("losses", [float(x) for x in out.raw["losses"]], "oracle total", float(total))
rows = []
for k, g in gref.items():
    h = got[k].cpu()
    rows.append((float((h - g).abs().max()) / (float(g.abs().max()) + 1e-30), k, float(g.abs().max())))
rows.sort(reverse=True)
rows = [r for r in rows if r[2] > 1e-6]
for r in rows[:2]:
    print("%.3e  %-60s scale %.3e" % r)
