"""Same-process A/B of the headline forward under two values of one environment setting that the model reads when it is built
(box-to-box and run-to-run spreads of +-3 % hide effects of 1-2 %; two processes in a row differ by the chip's temperature):
both models live in one process and take turns, ROUNDS x STEPS forwards each.
    python tools/ab_inproc.py VAR valueA valueB [rounds] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from helpers import make_inputs

var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda", 0)
hp = V.config("c2")
inputs, noise, _ = make_inputs(hp, seed=100, variant="A")
models = []
for val in (va, vb):
    if val == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = val
    m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device=dev)
    m.train(True)
    d = {}
    for k in ("traj_seq", "I_0", "I_g", "end_ind", "start_ind", "pad_mask", "traj_seq_states", "actions"):
        buf = m.input_buffer(k, inputs[k].shape) if k != "start_ind" else inputs[k].to(dev)
        buf.copy_(inputs[k])
        d[k] = buf
    for _ in range(3):
        m(d, "train")
    models.append((m, d))
torch.cuda.synchronize()
res = [[], []]
for r in range(rounds):
    order = (0, 1) if r % 2 == 0 else (1, 0)          # A B B A ...: whatever the position inside a round costs, both pay it equally
    for i in order:
        m, d = models[i]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = m(d, "train")
            m.loss(d, out)
        torch.cuda.synchronize()
        res[i].append(1e3 * (time.perf_counter() - t0) / steps)
for i, val in enumerate((va, vb)):
    v = sorted(res[i])
    print(f"{var}={val:8s} median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}   rounds: " + " ".join(f"{x:.3f}" for x in res[i]))
