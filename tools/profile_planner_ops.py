"""Count the torch / HIP operations of ONE CEM iteration on latents (torch.profiler with stacks): what the host side of the planner
launches besides the model's plan.  python tools/profile_planner_ops.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner
from torch.profiler import profile, ProfilerActivity
hp = V.config("c4")
m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda"); m.eval()
rng = np.random.RandomState(0)
state = rng.randint(0, 256, size=(1, 64, 64, 3)).astype(np.uint8); goal = rng.randint(0, 256, size=(1, 64, 64, 3)).astype(np.uint8)
sampler = SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=1)
planner = CEMPlanner(GCPImageSimulator(m, pred_length=False), LearnedCostEstimate(m), sampler, n_iters=1, batch_size=512, elite_frac=0.1, max_seq_len=80)
for _ in range(3): planner.iterate(state, goal)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    planner.iterate(state, goal); torch.cuda.synchronize()
rows = prof.key_averages(group_by_stack_n=6)
rows = sorted(rows, key=lambda r: -r.count)
for r in rows[:40]:
    st = [s for s in r.stack if "video-gcp_amd" in s or "video_gcp_amd" in s]
    print(f"{r.count:5d} {r.key[:50]:50s} {st[:2]}")
