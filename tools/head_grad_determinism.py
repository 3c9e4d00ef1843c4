"""Replays the training forward's output-head launch (conv3x3_head_split_kernel<2>: likelihood + its gradient rows) six times on the same
inputs and counts the gradient values that differ between replays (must be 0).  Round 4 found ~150 of 587 M values per launch coming out
as the register's previous content in lanes 32..63 (a VALU-result -> store-data hazard; csrc/conv3x3_split.hip, the s_nop in front of the
gradient-row stores): python tools/head_grad_determinism.py"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import video_gcp_amd as V
from video_gcp_amd import runtime as rt
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
hp = V.config("c2")
sd = V.init_params(hp, seed=3)
a = GCPTrainStep(GCPTreeModel(hp, params={k: v.clone() for k, v in sd.items()}, device="cuda"), lr=1e-3)
inputs, noise, _ = make_inputs(hp, seed=100, variant="A")
dev_in = {k: v.cuda() for k, v in inputs.items()}
nz = noise.cuda()
a.backward(dev_in, nz); torch.cuda.synchronize()
dMD = a.last_bplan.outs["dMD"]
m = a.m
plan = [v[1] for v in m._plans.values()][0]
heads = [(i, op) for i, op in enumerate(plan.ops) if "head" in op[0]]
print([op[0] for _, op in heads])
i, (name, fn, args, lane) = [(i, op) for i, op in heads if "dec.head" in op[0] or op[0].endswith("head")][-1]
print("replaying", name)
st = m._stream.cuda_stream
snaps = []
for r in range(6):
    rt.check(fn(*args, st), name)
    torch.cuda.synchronize()
    snaps.append(dMD.clone())
for r in range(1, 6):
    ne = snaps[0] != snaps[r]
    print("replay", r, "differing elements:", int(ne.sum()))
