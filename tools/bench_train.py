"""Per-op device time of the training step's backward plan (tuning aid): python tools/bench_train.py [config] [key=val]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import video_gcp_amd as V
from video_gcp_amd import runtime as rt
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    over = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[2:])}
    hp = V.config(name, **over)
    model = GCPTreeModel(hp, device="cuda")
    tr = GCPTrainStep(model)
    inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    for _ in range(3):
        tr.step(dev_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        tr.step(dev_in)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    print(f"training step: {ms:.2f} ms  = {hp.batch_size * hp.max_seq_len / ms * 1e3:.0f} frames/s (B={hp.batch_size})")
    t0 = time.perf_counter()
    for _ in range(K):
        model(dev_in, "train")
    torch.cuda.synchronize()
    print(f"  forward(+losses) only: {(time.perf_counter() - t0) / K * 1e3:.2f} ms")
    # per-op times of the backward plan (eager, event-bracketed)
    plan = tr.last_bplan
    res = {}
    st = model._stream
    with torch.cuda.stream(st):
        for nm, fn, args, _ in plan.ops:
            if nm.startswith("@"):
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            rt.check(fn(*args, st.cuda_stream), nm)
            e0.record(st)
            for _ in range(3):
                rt.check(fn(*args, st.cuda_stream), nm)
            e1.record(st)
            st.synchronize()
            res[nm] = res.get(nm, 0.0) + 1e3 * e0.elapsed_time(e1) / 3
    tot = sum(res.values())
    print(f"backward plan: {len(plan.ops)} launches, sum of op times {tot / 1e3:.2f} ms")
    for nm, us in sorted(res.items(), key=lambda kv: -kv[1])[:int(os.environ.get('TOP', '40'))]:
        print(f"  {nm:40s} {us:9.1f} us")
    groups = {}
    for nm, us in res.items():
        g = nm.split(":")[0]
        groups[g] = groups.get(g, 0.0) + us
    print("by kind:", {k: round(v) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])})


if __name__ == "__main__":
    main()
