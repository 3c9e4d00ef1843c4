import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for name in ("c2", "c5"):
    hp = V.config(name)
    t0 = time.perf_counter(); sd = V.init_params(hp, seed=0); t1 = time.perf_counter()
    m = GCPTreeModel(hp, params=sd, device="cuda"); torch.cuda.synchronize(); t2 = time.perf_counter()
    tr = GCPTrainStep(m); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{name}: init_params {t1-t0:.2f} s, model ctor {t2-t1:.2f} s, train-step ctor {t3-t2:.2f} s")
