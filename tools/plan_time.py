import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner
hp4 = V.config("c4")
m4 = GCPTreeModel(hp4, params=V.init_params(hp4, seed=0), device="cuda"); m4.eval()
rng = np.random.RandomState(0)
state = rng.randint(0, 256, size=(1, 64, 64, 3)).astype(np.uint8); goal = rng.randint(0, 256, size=(1, 64, 64, 3)).astype(np.uint8)
n = 512
sampler = SimpleTreeCEMSampler(float("inf"), None, hp4.nz_vae, 1.0, n_level_hierarchy=hp4.hierarchy_levels, device="cuda", seed=1)
sim, cost = GCPImageSimulator(m4, pred_length=False), LearnedCostEstimate(m4)
planner = CEMPlanner(sim, cost, sampler, n_iters=1, batch_size=n, elite_frac=0.1, max_seq_len=80)
def T(tag, fn, k=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = fn()
    torch.cuda.synchronize(); print(f"{tag:30s} {(time.perf_counter()-t0)/k*1e3:8.2f} ms"); return r
s = T("sample", lambda: sampler.sample(n))
r = T("rollout_device(decode=False)", lambda: sim.rollout_device(state, goal, s, 80, decode=False))
T("cost", lambda: cost.sequence_cost_device(r.latents, r.lengths, r.e_goal))
sc, _ = planner.evaluate(state, goal, s)
T("fit", lambda: sampler.fit(s[torch.argsort(sc)[: n // 10]]))
from video_gcp_amd.planning import env2planner
T("env2planner x2", lambda: (env2planner(np.repeat(state, n, 0)).cuda(), env2planner(np.repeat(goal, n, 0)).cuda()))
