"""Flat VRNN baseline (gcp_sequential) forward at c2 shapes: python tools/bench_sequential.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from helpers import make_inputs
hp = V.config("c2")
m = GCPSequentialModel(hp, device="cuda")
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
d = {k: v.cuda() for k, v in inputs.items() if k in ("traj_seq", "I_0", "I_g", "end_ind", "start_ind")}
nz = None
for _ in range(3):
    m(d, "train")
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
th = 0.0
for _ in range(K):
    h0 = time.perf_counter(); m(d, "train"); th += time.perf_counter() - h0
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
print(f"gcp_sequential forward (B={hp.batch_size}, T={hp.max_seq_len}): {ms:.2f} ms/step = {hp.batch_size * hp.max_seq_len / ms * 1e3:.0f} frames/s; host {th / K * 1e3:.2f} ms")
res = m.profile_ops(d, "train")
tot = sum(t for _, t in res)
groups = {}
for n, t in res:
    g = n.split(":")[0].rstrip("0123456789.")
    groups[g] = groups.get(g, 0) + t
print("ops", len(res), "sum", round(tot), "us; top groups:", {k: round(v) for k, v in sorted(groups.items(), key=lambda x: -x[1])[:12]})
