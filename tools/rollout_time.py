import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.planning import GCPImageSimulator
hp4 = V.config("c4")
m4 = GCPTreeModel(hp4, params=V.init_params(hp4, seed=0), device="cuda"); m4.eval()
rng = np.random.RandomState(0)
state = rng.randint(0, 256, size=(1, 64, 64, 3)).astype(np.uint8); goal = rng.randint(0, 256, size=(1, 64, 64, 3)).astype(np.uint8)
sim = GCPImageSimulator(m4, pred_length=False)
for n in (64, 512):
    s = np.random.RandomState(1).randn(n, hp4.n_nodes, hp4.nz_vae).astype(np.float32)
    sim.rollout(state, goal, s, 80); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = sim.rollout(state, goal, s, 80); torch.cuda.synchronize()
    print(f"rollout() n={n}: {(time.perf_counter()-t0)*1e3:.1f} ms; predictions[0] {r.predictions[0].shape}")
