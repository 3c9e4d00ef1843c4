import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs
from oracle import gcp_sequential_oracle as S
hp = V.config("c1", nz_mid_lstm=128, lstm_init="zero")
sd = V.init_params_sequential(hp, seed=1, randomize_affine=True)
model = GCPSequentialModel(hp, params=sd, device="cuda")
tr = SequentialTrainStep(model, lr=1e-3)
inputs, noise, _ = make_inputs(hp, seed=7, variant="B")
noise = noise[:, :hp.max_seq_len - 1].contiguous()
dev_in = {k: v.cuda() for k, v in inputs.items()}
for _ in range(2):
    out = tr.backward(dev_in, noise.cuda())
torch.cuda.synchronize()
gref, res, total, oref = S.gradients(sd, hp, inputs, noise)
got = tr.named_grads()
for k, g in gref.items():
    err, scale = float((got[k].cpu() - g).abs().max()), float(g.abs().max())
    flag = "BAD" if err > 1e-3 * scale + 5e-7 else "   "
    print(f"{flag} {k:60s} err {err:.3e} scale {scale:.3e} rel {err / (scale + 1e-30):.2e}")
