import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs
hp = V.config("c1", nz_mid_lstm=128, lstm_init="zero")
G = {"1": True, "0": False, "auto": "auto"}[os.environ.get("TT_GRAPH", "1")]
def mk():
    sd = V.init_params_sequential(hp, seed=1, randomize_affine=True)
    m = GCPSequentialModel(hp, params=sd, device="cuda")
    m.use_graph = G
    return m, SequentialTrainStep(m, lr=1e-3)
m1, t1 = mk(); m2, t2 = mk()
if os.environ.get("TT_SYNC_REPACK"):
    for m in (m1, m2):
        def wrap(orig):
            def f(*a, **k):
                mode = os.environ["TT_SYNC_REPACK"]
                if mode in ("1", "before"): torch.cuda.synchronize()
                r = orig(*a, **k)
                if mode in ("1", "after"): torch.cuda.synchronize()
                return r
            return f
        m.repack = wrap(m.repack)
if os.environ.get("TT_SYNC_FWD"):
    for m in (m1, m2):
        def wrapf(orig):
            def f(*a, **k):
                torch.cuda.synchronize(); r = orig(*a, **k); torch.cuda.synchronize(); return r
            return f
        m.forward = wrapf(m.forward)
for step in range(6):
    inputs, noise, _ = make_inputs(hp, seed=40 + step, variant="B")
    noise = noise[:, :hp.max_seq_len - 1].contiguous().cuda()
    dev = {k: v.cuda() for k, v in inputs.items()}
    o1 = t1.step(dev, noise)
    if os.environ.get("TT_SYNC") == "1": torch.cuda.synchronize()
    o2 = t2.step(dev, noise)
    torch.cuda.synchronize()
    g1, g2 = t1.named_grads(), t2.named_grads()
    bad = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    if bad and os.environ.get("TT_VERBOSE"):
        for k in bad:
            d = float((g1[k] - g2[k]).abs().max()); sc = float(g1[k].abs().max())
            print(f"      {k:60s} {d:.3e} / {sc:.3e}")
    def foldchk(m):
        from video_gcp_amd import packing as pk_
        out = []
        for net in m._nets:
            sd = m.sd; p = f"dense_rec.lstm.cell.{net}"
            Wih = sd[f"{p}.lstm.0.weight_ih"].double()
            w, b = pk_.lstm_gate_interleave((Wih @ sd[f"{p}.embed.weight"].double()).float(), sd[f"{p}.lstm.0.weight_hh"],
                                            (Wih @ sd[f"{p}.embed.bias"].double()).float() + sd[f"{p}.lstm.0.bias_ih"], sd[f"{p}.lstm.0.bias_hh"])
            out.append("%.1e" % float((m.pk[net]["lstm0f.w"] - pk_.pack_gemm(w)).abs().max()))
        return out
    print("      fold error m1", foldchk(m1), "m2", foldchk(m2), "packs equal", all(torch.equal(m1.pk[n][k], m2.pk[n][k]) for n in m1._nets for k in ("lstm0f.w", "lstm0f.b", "lstm1.w", "embed.w")))
    print(f"step {step}: losses equal {torch.equal(o1.raw['losses'], o2.raw['losses'])}, grads differ in {len(bad)} {bad[:4]}, theta equal {torch.equal(m1.theta, m2.theta)}", flush=True)

# ---- are each model's folded packs the fold of ITS parameters? ----
from video_gcp_amd import packing as pk
def expected(m, net):
    sd = m.sd
    p = f"dense_rec.lstm.cell.{net}"
    We, be = sd[f"{p}.embed.weight"].double(), sd[f"{p}.embed.bias"].double()
    Wih = sd[f"{p}.lstm.0.weight_ih"].double()
    w, b = pk.lstm_gate_interleave((Wih @ We).float(), sd[f"{p}.lstm.0.weight_hh"], (Wih @ be).float() + sd[f"{p}.lstm.0.bias_ih"], sd[f"{p}.lstm.0.bias_hh"])
    return pk.pack_gemm(w), b
for name, m in (("m1", m1), ("m2", m2)):
    for net in m._nets:
        w, b = expected(m, net)
        dw = float((m.pk[net]["lstm0f.w"] - w).abs().max()); db = float((m.pk[net]["lstm0f.b"] - b).abs().max())
        print(name, net, "fold vs parameters: max |dw| %.3e  max |db| %.3e" % (dw, db))
print("arena packs equal to a fresh gather:", [torch.equal(m.pk[n]["lstm1.w"], pk.pack_gemm(pk.lstm_gate_interleave(m.sd[f"dense_rec.lstm.cell.{n}.lstm.1.weight_ih"], m.sd[f"dense_rec.lstm.cell.{n}.lstm.1.weight_hh"], m.sd[f"dense_rec.lstm.cell.{n}.lstm.1.bias_ih"], m.sd[f"dense_rec.lstm.cell.{n}.lstm.1.bias_hh"])[0])) for m in (m1, m2) for n in m._nets])
