"""CEM planning rollout throughput (BASELINE configs[3]: 512 candidates x horizon 80, 25-room gcp_tree sizes, sharded over
the ranks that launch this script).  One "iteration" = sample -> rollout of this rank's candidate shard -> learned cost ->
all-gather -> elites -> refit.  Prints one JSON line (candidates/s and predicted frames/s, whole job)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import video_gcp_amd as V
from video_gcp_amd import dist as D
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner

ap = argparse.ArgumentParser()
ap.add_argument("--candidates", type=int, default=512)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
rank, local_rank, world = D.init_from_env()
torch.cuda.set_device(local_rank)
hp = V.config("c4")
model = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device=torch.device("cuda", local_rank))
model.eval()
rng = np.random.RandomState(0)
state = rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
goal = rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
sampler = SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=1)
planner = CEMPlanner(GCPImageSimulator(model, pred_length=False), LearnedCostEstimate(model), sampler, n_iters=1, batch_size=args.candidates,
                     elite_frac=0.1, max_seq_len=hp.max_seq_len)
for _ in range(2):
    s = sampler.sample(args.candidates)
    planner.evaluate(state, goal, s)
torch.cuda.synchronize(); D.barrier()
t0 = time.perf_counter()
for _ in range(args.iters):
    s = sampler.sample(args.candidates)
    scores, _ = planner.evaluate(state, goal, s)
    idx = torch.argsort(scores)[: args.candidates // 10]
    sampler.fit(s[idx])
torch.cuda.synchronize(); D.barrier()
dt = D.max_over_ranks(time.perf_counter() - t0, device="cuda")
if rank == 0:
    print(json.dumps({"workload": "CEM iteration, 512 candidates x horizon 80 (gcp_tree, 64x64, L=7), eval-mode BN, prior path with given z; candidates scored on latents (planner default: the decoder runs for the returned plan only)",
                      "n_gpus": world, "candidates_per_s": round(args.candidates * args.iters / dt, 1),
                      "predicted_frames_per_s": round(args.candidates * hp.max_seq_len * args.iters / dt, 1),
                      "ms_per_iteration": round(1e3 * dt / args.iters, 2), "candidates_per_gpu": args.candidates // world}))
