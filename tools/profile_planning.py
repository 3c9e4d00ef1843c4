"""Per-op device times of the planner's scoring rollout (c4 model, B candidates, given z, eval BN): python tools/profile_planning.py [B] [decode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
decode = len(sys.argv) > 2 and sys.argv[2] == "1"
hp = V.config("c4")
m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda")
m.eval()
g = torch.Generator().manual_seed(0)
inp = dict(I_0=torch.rand(B, 3, 64, 64, generator=g).cuda() * 2 - 1, I_g=torch.rand(B, 3, 64, 64, generator=g).cuda() * 2 - 1,
           z=torch.randn(B, hp.n_nodes, hp.nz_vae, generator=g).cuda(), end_ind=torch.full((B,), hp.max_seq_len - 1, dtype=torch.long).cuda())
with m.val_mode(pred_length=False, decode=decode):
    res = m.profile_ops(inp, "train")
tot = sum(t for _, t in res)
for n, t in sorted(res, key=lambda x: -x[1])[:28]:
    print(f"{n:28s} {t:9.1f} us")
print("total", round(tot), "us over", len(res), "ops")
