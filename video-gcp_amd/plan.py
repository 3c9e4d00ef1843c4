"""Planning entry point: `python -m video_gcp_amd.plan --config c4 --weights <ckpt> --nstart_goal_pairs K [--planner cem|hierarchical]`.

Counterpart of the device path behind /root/reference/gcp/planning/run.py: for every (start, goal) image pair the planner
(`CEMPlanner` / `HierarchicalCEMPlanner`, gcp/planning/cem/cem_planner.py:55-218) searches the latent space of the
goal-conditioned predictor and returns (image_plan, actions, latents, cost) (cem_planner.py:96).  What run.py adds around
that — the MiniWorld simulator, agents, policies, trajectory savers (gcp/planning/infra/*) — is host infrastructure outside
SURVEY.md §8 and is not rebuilt: start / goal frames come from an .npz (`--pairs file.npz` with uint8 arrays `start`,
`goal` [K, H, W, 3]) or are seeded synthetic images.  Two ways to use several GPUs, as in the reference and in SURVEY §8(e):
  * one process per GPU over disjoint pair ranges (run.py:78-94 splits start_index..end_index over workers) — `--split pairs`;
  * one process per GPU sharing every CEM population (candidates sharded, one all-gather of costs per iteration) —
    `--split candidates` (the default under torch.distributed.run)."""
import argparse
import json
import os

import numpy as np
import torch


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="c4")
    p.add_argument("--weights", default="", help="checkpoint (weights_ep*.pth); random init when empty")
    p.add_argument("--pairs", default="", help=".npz with uint8 `start`, `goal` [K, H, W, 3]")
    p.add_argument("--nstart_goal_pairs", type=int, default=4)
    p.add_argument("--planner", default="cem", choices=["cem", "hierarchical"])
    p.add_argument("--candidates", type=int, default=512)
    p.add_argument("--iters", type=int, default=3)
    p.add_argument("--elite_frac", type=float, default=0.1)
    p.add_argument("--split", default="candidates", choices=["candidates", "pairs"])
    p.add_argument("--out", default="", help="directory for plan_{i}.npz")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--reference_rng", action="store_true",
                   help="hierarchical planner: replay the reference's np.random call sequence draw for draw (6x the host time per "
                        "call) instead of drawing only the kept rows from a seeded numpy Generator")
    return p.parse_args(argv)


def main(argv=None):
    from . import dist as D
    from .checkpoint import load_weights
    from .hparams import config
    from .model import GCPTreeModel
    from .params import init_params
    from . import planning as P
    args = get_args(argv)
    rank, local_rank, world = D.init_from_env()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    hp = config(args.config)
    model = GCPTreeModel(hp, params=init_params(hp, seed=args.seed), device=dev)
    if args.weights:
        load_weights(args.weights, model)
    model.eval()                                                     # planner_policy.py:51
    if args.pairs:
        z = np.load(args.pairs)
        starts, goals = z["start"], z["goal"]
    else:
        rng = np.random.RandomState(args.seed)
        K = args.nstart_goal_pairs
        starts = rng.randint(0, 256, size=(K, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
        goals = rng.randint(0, 256, size=(K, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
    idx = list(range(len(starts)))
    if args.split == "pairs" and world > 1:                          # run.py:78-94: disjoint index ranges per worker
        per = (len(idx) + world - 1) // world
        idx = idx[rank * per:(rank + 1) * per]
        if torch.distributed.is_initialized():                       # no collective on this path: plan alone
            torch.distributed.destroy_process_group()
    sim, cost = P.GCPImageSimulator(model), P.LearnedCostEstimate(model)
    if args.planner == "cem":
        sampler = P.SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device=dev,
                                         seed=args.seed + 1)
        planner = P.CEMPlanner(sim, cost, sampler, n_iters=args.iters, batch_size=args.candidates, elite_frac=args.elite_frac,
                               max_seq_len=hp.max_seq_len)
    else:
        planner = P.HierarchicalCEMPlanner(sim, cost, hp.hierarchy_levels, [10, 10], action_dim=hp.nz_vae,       # rates of the
                                           max_seq_len=hp.max_seq_len,                                       # 25-room control config
                                           fast_draws=not args.reference_rng, seed=args.seed + 1)
    results = []
    for i in idx:
        plan, actions, latents, c = planner(starts[i:i + 1], goals[i:i + 1])
        results.append({"pair": int(i), "cost": float(c), "plan_len": int(plan.shape[0])})
        if args.out and (rank == 0 or args.split == "pairs"):
            os.makedirs(args.out, exist_ok=True)
            np.savez_compressed(os.path.join(args.out, f"plan_{i}.npz"), image_plan=plan, latents=latents,
                                actions=(actions if actions is not None else np.zeros(0)), cost=c)
    if rank == 0 or args.split == "pairs":
        print(json.dumps({"rank": rank, "plans": results}), flush=True)
    return results


if __name__ == "__main__":
    main()
