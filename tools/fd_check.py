"""Diagnostic: directional finite differences of the HIP loss vs <grad, d> for sub-groups of the conv encoder at c2 / B=16."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hp = V.config("c3", batch_size=B)
sd = V.init_params(hp, seed=1, randomize_affine=True)
model = GCPTreeModel(hp, params=sd, device="cuda")
tr = GCPTrainStep(model, lr=1e-3)
inputs, noise, _ = make_inputs(hp, seed=9, variant="B")
dev_in = {k: v.cuda() for k, v in inputs.items()}
dnoise = noise.cuda()
out = tr.backward(dev_in, dnoise)
torch.cuda.synchronize()
grad = tr.grad.clone(); theta0 = model.theta.clone()
div = float(hp.max_seq_len * hp.input_nc * hp.img_sz ** 2)
def terms_at(theta):
    model.theta.copy_(theta); model.repack()
    o = model(dev_in, "train", noise=dnoise); torch.cuda.synchronize()
    return o.raw["losses"].double().cpu()
gen = torch.Generator(device="cuda").manual_seed(0)
for pre in ["encoder.net.input", "encoder.net.pyramid-0.conv", "encoder.net.pyramid-0.norm", "encoder.net.pyramid-1", "encoder.net.pyramid-2", "encoder.net.head", "encoder.", "inf_encoder."]:
    mask = torch.zeros_like(theta0)
    for k, (o, shp) in model._poff.items():
        if k.startswith(pre) and not k.endswith(("running_mean", "running_var")):
            mask[o:o + int(np.prod(shp))] = 1.0
    g = grad * mask
    d = g / g.norm().clamp_min(1e-30)
    analytic = float((grad.double() * d.double()).sum())
    scale = float(theta0[mask > 0].abs().mean()) / max(float(d.abs().max()), 1e-12)
    row = [f"{pre:30s} analytic {analytic:.6f}"]
    for rel in (4e-3, 2e-3, 1e-3, 5e-4):
        eps = rel * scale
        tp, tm = terms_at(theta0 + eps * d), terms_at(theta0 - eps * d)
        fds = [float(tp[i] - tm[i]) / div / (2 * eps) for i in range(4)]
        row.append(f"rel {rel:g}: fd {sum(fds):.6f} (rec {fds[0]:.6f} kl {fds[1]:.6f} len {fds[2]:.2e} ex {fds[3]:.2e})")
    print("\n   ".join(row), flush=True)
