"""Gradient parity of the HIP training step against torch autograd over the oracle (debug driver; the pytest version is
tests/test_gpu_training.py).  usage: python tools/check_grads.py [config] [key=value ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
from oracle import gcp_model_oracle as O


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    over = {}
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        over[k] = int(v)
    hp = V.config(name, **over)
    sd = V.init_params(hp, seed=1, randomize_affine=True)
    model = GCPTreeModel(hp, params=sd, device="cuda")
    model.use_graph = os.environ.get("GRAPH", "0") == "1"
    tr = GCPTrainStep(model)
    inputs, noise, _ = make_inputs(hp, seed=7, variant="B")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    out = tr.backward(dev_in, noise.cuda())
    torch.cuda.synchronize()
    taps = {}
    gref, res, total, ref = O.gradients(sd, hp, inputs, noise, taps)
    if os.environ.get("TAPS"):
        bo = tr.last_bplan.outs
        L, N, nz, nv = hp.hierarchy_levels, hp.n_nodes, hp.nz_enc, hp.nz_vae
        def cmp(name, got, want):
            got, want = got.detach().float().cpu(), want.detach().float().cpu()
            print(f"  tap {name:28s} err {float((got - want).abs().max()):.3e} scale {float(want.abs().max()):.3e}")
        dE, dHid = bo["dE"], bo["dHid"]
        bf2df = out.tree._bf2df.cpu()
        dbf = taps["bf_e_g_prime"].grad                      # decoder + existence part, bf order
        got = (bo["dE_dec"] + bo["dE_ex"]).view(hp.batch_size, N, nz).cpu()[:, bf2df]
        cmp("dE_dec+dE_ex", got, dbf)
        for l in range(L):
            s_, n = 2 ** (L - 1 - l), 2 ** l
            slots = torch.tensor([(2 * j + 1) * s_ for j in range(n)])
            cmp(f"dE level {l}", dE.cpu()[:, slots], taps[f"e_g_prime.{l}"].grad)
            cmp(f"dHid level {l}", dHid.cpu()[:, slots], taps[f"hidden.{l}"].grad) if taps[f"hidden.{l}"].grad is not None else None
            cmp(f"dq_mu level {l}", bo["dQZ"].cpu()[:, slots][..., :nv], taps[f"q_z_mu.{l}"].grad) if False else None
        cmp("dlen", bo["dlen"][:, :hp.max_seq_len], taps["seq_len_logits"].grad)
        print(bo["dlen"].cpu(), taps["seq_len_logits"].grad)
        cmp("dE slot0 (e_0)", dE.cpu()[:, 0], taps["e_0"].grad)
        cmp("dE slot2^L (e_g)", dE.cpu()[:, 2 ** L], taps["e_g"].grad)
        cmp("d inf_enc_seq", bo["d_inf"].view(hp.batch_size, hp.max_seq_len, nz), taps["inf_enc_seq"].grad)
        cmp("d enc_traj_seq", bo["d_enc_traj"].view(hp.batch_size, hp.max_seq_len, nz), taps["enc_traj_seq"].grad)
        md = taps["matched_distr"].grad                       # [B,T,100,S,S] canonical order
        perm = model._dlm_perm.cpu()
        g = bo["dMD"].view(hp.batch_size, hp.max_seq_len, hp.img_sz, hp.img_sz, -1).cpu()
        inv = torch.empty(hp.head_channels, dtype=torch.long)
        slots = torch.nonzero(perm >= 0)[:, 0]
        inv[perm[slots]] = slots
        cmp("d matched_distr", g.index_select(-1, inv).permute(0, 1, 4, 2, 3), md)
    print("total", float(total), "hip", float(out.raw["losses"][5]))
    got = tr.named_grads()
    worst = []
    for k, g in gref.items():
        h = got[k].cpu()
        err = float((h - g).abs().max())
        scale = float(g.abs().max())
        rel = err / (scale + 1e-12)
        worst.append((rel if scale > 1e-9 else err, k, err, scale))
    worst.sort(reverse=True)
    nbad = 0
    for rel, k, err, scale in worst:
        bad = (err > 2e-3 * scale + 1e-9)
        nbad += bad
        if bad or os.environ.get("ALL"):
            print(f"{'BAD ' if bad else 'ok  '}{k:75s} err {err:.3e} scale {scale:.3e}")
    print(f"{len(worst) - nbad}/{len(worst)} parameter gradients match")
    for rel, k, err, scale in worst[:8]:
        print(f"   worst: {k:70s} err {err:.3e} scale {scale:.3e}")


if __name__ == "__main__":
    main()
