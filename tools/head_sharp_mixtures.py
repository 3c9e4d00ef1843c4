"""Headline forward (+ losses) and training step with SHARP mixtures: the head's log-scale biases set to -6, as in a trained model, so that
most (mixture, colour) bins are vanishing and the likelihood's fallback branch (log-density at the bin centre) runs for nearly every
wavefront instead of never (random-init weights).  python tools/head_sharp_mixtures.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
hp = V.config("c2")
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")


def run(ls_bias):
    sd = V.init_params(hp, seed=0)
    if ls_bias is not None:
        nm = hp.n_mixtures
        b = sd["decoder.gen_head.conv.bias"]
        for c in range(3):                                  # canonical order: [logits | per colour: means, log_scales, coeffs]
            b[nm + c * 3 * nm + nm: nm + c * 3 * nm + 2 * nm] = ls_bias
    model = GCPTreeModel(hp, params=sd, device="cuda")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    nz = noise.cuda()
    for _ in range(3):
        out = model(dev_in, "train", noise=nz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = model(dev_in, "train", noise=nz)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / 20 * 1e3
    loss = float(out.raw["losses"][5])
    tr = GCPTrainStep(model)
    for _ in range(3):
        tr.step(dev_in, nz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        tr.step(dev_in, nz)
    torch.cuda.synchronize()
    return fwd, (time.perf_counter() - t0) / 8 * 1e3, loss


for name, v in (("random init (log-scales ~ 0)", None), ("log-scale bias -3", -3.0), ("log-scale bias -6", -6.0)):
    f, t, l = run(v)
    print(f"{name:32s} forward + losses {f:.3f} ms   training step {t:.2f} ms   total loss {l:.3f}")
