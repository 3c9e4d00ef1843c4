"""Time per (start, goal) pair through the planning entry point's loop (CEM, 512 candidates x 3 iterations, c4 model)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd import planning as P
hp = V.config("c4")
m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda"); m.eval()
rng = np.random.RandomState(0)
starts = rng.randint(0, 256, size=(6, 64, 64, 3)).astype(np.uint8); goals = rng.randint(0, 256, size=(6, 64, 64, 3)).astype(np.uint8)
sim, cost = P.GCPImageSimulator(m), P.LearnedCostEstimate(m)
sampler = P.SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=1)
planner = P.CEMPlanner(sim, cost, sampler, n_iters=3, batch_size=512, elite_frac=0.1, max_seq_len=hp.max_seq_len)
planner(starts[:1], goals[:1]); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, 6):
    plan, actions, lat, c = planner(starts[i:i + 1], goals[i:i + 1])
torch.cuda.synchronize()
print(f"CEM plan (512 candidates x 3 iterations + final decode): {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per pair; plan {plan.shape}")
