"""Two trainers of the flat model on the same minibatches: where do they first differ?  (env switches select the schedule)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs
hp = V.config("c1", nz_mid_lstm=128, lstm_init="zero")
def mk():
    sd = V.init_params_sequential(hp, seed=1, randomize_affine=True)
    m = GCPSequentialModel(hp, params=sd, device="cuda")
    return m, SequentialTrainStep(m, lr=1e-3)
m1, t1 = mk(); m2, t2 = mk()
print("start equal:", torch.equal(m1.theta, m2.theta))
for step in range(4):
    inputs, noise, _ = make_inputs(hp, seed=40 + step, variant="B")
    noise = noise[:, :hp.max_seq_len - 1].contiguous().cuda()
    dev = {k: v.cuda() for k, v in inputs.items()}
    SYNC = os.environ.get("TT_SYNC", "1") == "1"
    o1 = t1.backward(dev, noise)
    if SYNC: torch.cuda.synchronize()
    o2 = t2.backward(dev, noise); torch.cuda.synchronize()
    ge = torch.equal(t1.grad, t2.grad)
    le = torch.equal(o1.raw["losses"], o2.raw["losses"])
    if not ge:
        g1, g2 = t1.named_grads(), t2.named_grads()
        bad = [k for k in g1 if not torch.equal(g1[k], g2[k])]
        print("   grads differ in", len(bad), "of", len(g1), bad[:6])
    t1.optimizer_step()
    if SYNC: torch.cuda.synchronize()
    t2.optimizer_step(); torch.cuda.synchronize()
    te = torch.equal(m1.theta, m2.theta)
    fe = all(torch.equal(m1.pk[n][k], m2.pk[n][k]) for n in m1._nets for k in m1.pk[n] if k in m2.pk[n])
    print(f"step {step}: losses equal {le}, grads equal {ge}, theta equal {te}, packs equal {fe}")
