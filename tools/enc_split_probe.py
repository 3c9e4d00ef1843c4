"""Encoder activations with and without the split-f16 encoder kernels, c1 shapes (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from helpers import make_inputs
hp = V.config("c1", batch_size=3)
sd = V.init_params(hp, seed=1, randomize_affine=True)
inputs, noise, _ = make_inputs(hp, seed=3, variant="A")
model = GCPTreeModel(hp, params=sd, device="cuda")
model.use_graph = False
dev_in = {k: v.cuda() for k, v in inputs.items()}
snap = {}
for mode in ("split", "exact"):
    if mode == "exact":
        os.environ["GCPX_ENC_NOSPLIT"] = "1"
    out = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    snap[mode] = {k: v.clone() for k, v in model._bufs.items() if v.dtype == torch.float32} if hasattr(model, "_bufs") else {}
if not snap["split"]:
    print([a for a in dir(model) if "buf" in a.lower()])
rows = []
for k, v in snap["split"].items():
    w = snap["exact"][k]
    d = float((v - w).abs().max())
    s = float(w.abs().max())
    if d > 0:
        rows.append((d / (s + 1e-30), str(k[0]), d, s, tuple(v.shape)))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%.3e %-28s abs %.3e scale %.3e %s" % r)
