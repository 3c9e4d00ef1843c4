"""Exercise the N > 1 code paths of bench.py on ONE GPU: a 1-rank "nccl" (= RCCL) process group, the barrier / max-reduce timing rule,
the gradient all-reduce inside the training step and the CEM cost all-gather.  Checks API usage on real RCCL, not scaling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")   # overridden by the test
import numpy as np
import torch
import torch.distributed as dist
import video_gcp_amd as V
from video_gcp_amd import dist as D
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
hp = V.config("c1")
m = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device=dev)
inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
d = {k: v.to(dev) for k, v in inputs.items()}
dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter(); m(d, "train", noise=noise.to(dev)); torch.cuda.synchronize(); dist.barrier()
print("forward + barrier ok; max over ranks:", D.max_over_ranks(time.perf_counter() - t0, device=dev))
tr = GCPTrainStep(m, process_group=dist.group.WORLD)
ref = GCPTrainStep(GCPTreeModel(hp, params=V.init_params(hp, seed=0), device=dev))
for _ in range(2):
    tr.step(d, noise.to(dev)); ref.step(d, noise.to(dev))
torch.cuda.synchronize()
err = float((tr.m.theta - ref.m.theta).abs().max())
print("training step with RCCL all-reduce (1 rank) vs no process group: max |theta diff| =", err)
assert err == 0.0
# the per-bucket timing bench.py --gpus N reports (event pairs on the communication stream around every bucket's all-reduce)
tr.buckets.timing = True
tr.step(d, noise.to(dev)); ref.step(d, noise.to(dev))
torch.cuda.synchronize()
per = tr.buckets.comm_ms()
tr.buckets.timing = False
assert len(per) == len(tr.buckets.ranges) and all(v >= 0.0 for v in per.values()), per
assert float((tr.m.theta - ref.m.theta).abs().max()) == 0.0
print("bucket all-reduce times (ms):", {k: round(v, 3) for k, v in per.items()})
from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner
m.eval()
rng = np.random.RandomState(0)
state = rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8); goal = rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
sampler = SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device=dev, seed=1)
planner = CEMPlanner(GCPImageSimulator(m), LearnedCostEstimate(m), sampler, n_iters=1, batch_size=16, elite_frac=0.25, max_seq_len=hp.max_seq_len)
plan, actions, lat, score = planner(state, goal)
print("CEM with RCCL all-gather (1 rank): plan", plan.shape, "score", score)
dist.destroy_process_group()
print("ok")
