"""Adaptive-binding forward (BASELINE configs[4] per-GPU shard: 64x64, T=200, L=8, 255 nodes, batch 8) — step time and the
per-op device times of the plan:  python tools/bench_adaptive.py [key=val ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from helpers import make_inputs


def main():
    over = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:])}
    hp = V.config("c5", **over)
    model = GCPTreeModel(hp, device="cuda")
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    nz = noise.cuda()
    for _ in range(3):
        out = model(dev_in, "train", noise=nz)
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        out = model(dev_in, "train", noise=nz)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    print(f"adaptive forward + losses (B={hp.batch_size}, T={hp.max_seq_len}, N={hp.n_nodes}): {ms:.2f} ms/step = "
          f"{hp.batch_size * hp.max_seq_len / ms * 1e3:.0f} predicted frames/s; total loss {float(out.raw['losses'][5]):.4f}")
    res = model.profile_ops(dev_in, "train", nz)
    tot = sum(us for _, us in res)
    print(f"{len(res)} launches, sum of op times {tot / 1e3:.2f} ms")
    for nm, us in sorted(res, key=lambda kv: -kv[1])[:30]:
        print(f"  {nm:32s} {us:9.1f} us")
    groups = {}
    for nm, us in res:
        g = nm.split(":")[0].rstrip("0123456789").split(".")[0]
        groups[g] = groups.get(g, 0.0) + us
    print("by kind:", {k: round(v) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])})


if __name__ == "__main__":
    main()
