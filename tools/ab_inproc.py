"""Same-process A/B of the headline forward under two values of one environment setting that the model reads when it is built
(box-to-box and run-to-run spreads of +-3 % hide effects of 1-2 %; two processes in a row differ by the chip's temperature):
two models per value live in one process (built A B B A) and take turns, ROUNDS x STEPS forwards each.
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
# FOUR instances, built A B B A: a model built earlier in the process runs 0.02-0.08 ms slower than one built later (measured with two
# identical settings), so each setting gets one early and one late instance
for val in (va, vb, vb, va):
    if var in ("@stream", "@graph"):           # not environment settings: which stream the CALLER works on ("caller" = the current stream,
        pass                                   # "model" = the model's own); hipGraph replay ("1") or eager launches of the plan ("0")
    elif val == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = val
    m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device=dev)
    m.train(True)
    if var == "@graph":
        m.use_graph = {"1": True, "0": False}.get(val, "auto")
    d = {}
    for k in ("traj_seq", "I_0", "I_g", "end_ind", "start_ind", "pad_mask", "traj_seq_states", "actions"):
        buf = m.input_buffer(k, inputs[k].shape) if k != "start_ind" else inputs[k].to(dev)
        buf.copy_(inputs[k])
        d[k] = buf
    for _ in range(3):
        m(d, "train")
    models.append((m, d))
torch.cuda.synchronize()
print("plans replayed eagerly:", [[p.eager for _, p in m._plans.values()] for m, _ in models], " use_graph:", [m.use_graph for m, _ in models],
      " tuned (graph, eager) ms:", [[tuple(round(1e3 * t, 3) for t in getattr(p, "tuned", (0, 0))) for _, p in m._plans.values()] for m, _ in models])
res = [[], [], [], []]
vals4 = (va, vb, vb, va)
for r in range(rounds):
    order = (0, 1, 2, 3) if r % 2 == 0 else (3, 2, 1, 0)     # whatever the position inside a round costs, both settings pay it equally
    for i in order:
        m, d = models[i]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(m._stream if (var == "@stream" and vals4[i] == "model") else torch.cuda.current_stream(dev)):
            for _ in range(steps):
                out = m(d, "train")
                m.loss(d, out)
        torch.cuda.synchronize()
        res[i].append(1e3 * (time.perf_counter() - t0) / steps)
med = lambda t: sorted(t)[len(t) // 2]
for val, idx in ((va, (0, 3)), (vb, (1, 2))):
    both = res[idx[0]] + res[idx[1]]
    print(f"{var}={val:8s} mean of the two instances' medians {(med(res[idx[0]]) + med(res[idx[1]])) / 2:.3f} ms  "
          f"(instance built {idx[0] + 1}.: {med(res[idx[0]]):.3f}, built {idx[1] + 1}.: {med(res[idx[1]]):.3f})  min {min(both):.3f}  max {max(both):.3f}")
