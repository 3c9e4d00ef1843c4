"""cProfile of one HierarchicalCEMPlanner call at the bench's setting (c4 model, rates [10, 10]): where the host time goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, HierarchicalCEMPlanner
hp = V.config("c4")
m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device="cuda")
m.eval()
rng = np.random.RandomState(0)
state = rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
goal = rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
# FAST=1: fast_draws (numpy Generator, kept rows only); default: the reference's np.random stream
pl = HierarchicalCEMPlanner(GCPImageSimulator(m, pred_length=False), LearnedCostEstimate(m), hp.hierarchy_levels, [10, 10], action_dim=hp.nz_vae,
                            max_seq_len=hp.max_seq_len, fast_draws=bool(os.environ.get("FAST")))
for _ in range(3):
    pl(state, goal)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    pl(state, goal)
torch.cuda.synchronize()
print("ms per call: %.2f" % (1e3 * (time.perf_counter() - t0) / 5))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    pl(state, goal)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
