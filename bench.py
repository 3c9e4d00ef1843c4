#!/usr/bin/env python
"""bench.py — predicted frames/sec of the gcp_tree hot path on MI355X (BASELINE.json metric).

One "step" = one forward pass of the goal-conditioned hierarchical predictor over one synthetic batch of
BASELINE.json configs[1] (25-room gcp_tree, 64x64x3, seq_len 80, batch 16 per GPU): encoder over B*T frames,
L=7 tree levels (posterior path), decoder over all B*127 tree nodes, balanced binding + matched/pruned gathers
and the auxiliary heads — everything `model(inputs)` runs (train.py:205-214 / cem_simulator.py:29-31).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HEAD_TRAFFIC_BYTES = int((2 * 328209.2 + 97536.0) * 1024)   # measured, see "traffic" below
F32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CUs @ 2.4 GHz


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (a port: the reference itself cannot run) on a bounded sample of the same
    workload: c2 shapes with batch 2, timed on the host cores."""
    import torch
    import video_gcp_amd as V
    from oracle import gcp_model_oracle as O
    from helpers import make_inputs
    hp = V.config("c2", batch_size=2)
    sd = V.init_params(hp, seed=0)
    inputs, noise, _ = make_inputs(hp, seed=0, variant="A")
    with torch.no_grad():
        O.forward(sd, hp, inputs, noise=noise, training_bn=True)      # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            O.forward(sd, hp, inputs, noise=noise, training_bn=True)
            n += 1
            dt = time.perf_counter() - t0
            if dt > seconds_budget or n >= 10:
                break
    fps = n * hp.batch_size * hp.max_seq_len / dt
    return {"value": round(fps, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} forward passes of oracle/gcp_model_oracle.py at c2 shapes with batch 2 "
                      f"(64x64, T=80, 127 nodes/seq), torch {torch.__version__} CPU fp32, {dt:.1f} s"}


def _timed(fn, steps, warmup, world, dev):
    """barrier + sync on both sides, max over ranks (same rule as the headline measurement)"""
    import torch
    import torch.distributed as dist
    from video_gcp_amd import dist as D
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return D.max_over_ranks(time.perf_counter() - t0, device=dev) / steps


def extras(args, model, hp, dinp, dnoise, inputs, rank, world, dev):
    """Secondary measurements of SURVEY.md §8(d): (iii) training step (forward + backward + RAdam, RCCL all-reduce of the
    flat gradient when N > 1), (ii) one CEM planning iteration over 512 candidates sharded over the ranks, and the
    adaptive-binding forward of configs[4].  Each is whole-job predicted frames/s; failures are reported, not hidden."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    from helpers import make_inputs
    res = {}
    k = max(3, min(args.steps, 10))
    model.set_timed_op(None)
    try:
        from video_gcp_amd.training import GCPTrainStep
        tr = GCPTrainStep(model, process_group=(dist.group.WORLD if world > 1 else None))
        full = {k_: v.to(dev) for k_, v in inputs.items()}
        dt = _timed(lambda: tr.step(full, dnoise), k, 2, world, dev)
        res["train_step"] = {"value": round(world * hp.batch_size * hp.max_seq_len / dt, 1), "unit": "frames/s",
                             "ms_per_step": round(1e3 * dt, 3), "workload": "configs[2] shard: forward + ELBO losses + backward + "
                             "RAdam, batch 16/GPU" + (", one RCCL all-reduce of the flat fp32 gradient per step" if world > 1 else ""),
                             "grad_mbytes": round(tr.grad.numel() * 4 / 1e6, 1)}
        del tr
    except Exception as e:  # noqa: BLE001
        res["train_step"] = {"error": repr(e)[:300]}
    try:
        from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner
        hp4 = V.config("c4")
        m4 = GCPTreeModel(hp4, params=V.init_params(hp4, seed=0), device=dev)
        m4.eval()
        rng = np.random.RandomState(0)
        state = rng.randint(0, 256, size=(1, hp4.img_sz, hp4.img_sz, 3)).astype(np.uint8)
        goal = rng.randint(0, 256, size=(1, hp4.img_sz, hp4.img_sz, 3)).astype(np.uint8)
        n = 512
        sampler = SimpleTreeCEMSampler(float("inf"), None, hp4.nz_vae, 1.0, n_level_hierarchy=hp4.hierarchy_levels, device=dev, seed=1)
        planner = CEMPlanner(GCPImageSimulator(m4), LearnedCostEstimate(m4), sampler, n_iters=1, batch_size=n, elite_frac=0.1,
                             max_seq_len=hp4.max_seq_len)

        def it():
            s = sampler.sample(n)
            scores, _ = planner.evaluate(state, goal, s)
            sampler.fit(s[torch.argsort(scores)[: n // 10]])
        # reference-equivalent work first: every candidate's 80 frames are decoded while scoring (cem_simulator.py:29-59) ...
        planner.decode_candidates = True
        dt_all = _timed(it, 3, 1, world, dev)
        # ... then the default: the learned cost reads latents only, so scoring skips the decoder (same scores, elites and plan;
        # tests/test_gpu_planning.py) and only the returned plan is decoded
        planner.decode_candidates = False
        dt = _timed(it, 3, 1, world, dev)
        res["planning_iteration"] = {"value": round(n * hp4.max_seq_len / dt_all, 1), "unit": "frames/s",
                                     "ms_per_iteration": round(1e3 * dt_all, 2), "candidates_per_s": round(n / dt_all, 1),
                                     "workload": "configs[3]: one CEM iteration, 512 candidates x horizon 80 sharded over the ranks, "
                                     "every candidate decoded (reference-equivalent work), rollout + learned cost on device, one "
                                     "all-gather of costs",
                                     "latent_scoring": {"ms_per_iteration": round(1e3 * dt, 2), "candidates_per_s": round(n / dt, 1),
                                                        "note": "planner default: candidates scored on latents, decoder skipped while "
                                                                "scoring (bit-identical scores / elites / plan); only the returned plan "
                                                                "is decoded"}}
        del planner, m4
    except Exception as e:  # noqa: BLE001
        res["planning_iteration"] = {"error": repr(e)[:300]}
    try:
        hp5 = V.config("c5")
        m5 = GCPTreeModel(hp5, params=V.init_params(hp5, seed=0), device=dev)
        i5, n5, _ = make_inputs(hp5, seed=200 + rank, variant="A")
        d5 = {k_: v.to(dev) for k_, v in i5.items()}
        n5 = n5.to(dev)
        dt = _timed(lambda: m5(d5, "train", noise=n5), k, 2, world, dev)
        res["adaptive_forward"] = {"value": round(world * hp5.batch_size * hp5.max_seq_len / dt, 1), "unit": "frames/s",
                                   "ms_per_step": round(1e3 * dt, 3), "workload": "configs[4] shard: adaptive (soft-DTW) binding + "
                                   "attentive inference forward with losses, 64x64, seq_len 200, L=8 (255 nodes), batch 8/GPU"}
        from video_gcp_amd.training import GCPTrainStep
        tr5 = GCPTrainStep(m5, process_group=(dist.group.WORLD if world > 1 else None))
        dt = _timed(lambda: tr5.step(d5, n5), max(3, k // 2), 2, world, dev)
        res["adaptive_train_step"] = {"value": round(world * hp5.batch_size * hp5.max_seq_len / dt, 1), "unit": "frames/s",
                                      "ms_per_step": round(1e3 * dt, 3), "workload": "configs[4] shard: forward + losses + backward "
                                      "+ RAdam of the adaptive model, batch 8/GPU" + (", RCCL all-reduce of the flat gradient" if world > 1 else "")}
        del tr5, m5
    except Exception as e:  # noqa: BLE001
        res["adaptive_forward"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="sequences per GPU (configs[1]: 16)")
    ap.add_argument("--eval-bn", action="store_true", help="running-stat BatchNorm (planner mode) instead of batch stats")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (training step, CEM "
                    "planning iteration, adaptive-binding forward) reported under \"also\"")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    from helpers import make_inputs

    from video_gcp_amd import dist as D
    if args.gpus > 1:
        assert int(os.environ.get("WORLD_SIZE", "1")) == args.gpus, \
            f"launch with python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}"
        rank, local_rank, world = D.init_from_env("nccl")      # "nccl" is RCCL on ROCm
    else:
        rank, local_rank, world = 0, 0, 1
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    hp = V.config("c2", batch_size=args.batch)
    model = GCPTreeModel(hp, params=V.init_params(hp, seed=0), device=dev)      # same seed on every rank = replicas
    model.train(not args.eval_bn)
    # the path shards by sequence: every rank predicts its own batch of independent sequences (weak scaling), no
    # data-path collective in the forward
    inputs, noise, _ = make_inputs(hp, seed=D.shard_seed(100, rank), variant="A")
    # headline = pure prediction forward: without pad_mask the model does not run its loss kernels
    dinp = {k: v.to(dev) for k, v in inputs.items() if k in ("traj_seq", "I_0", "I_g", "end_ind", "start_ind")}
    dnoise = noise.to(dev)

    for _ in range(max(args.warmup, 1)):
        model(dinp, "train", noise=dnoise)
    torch.cuda.synchronize()

    model.set_timed_op("dec.head")
    model(dinp, "train", noise=dnoise)          # builds the split graphs
    torch.cuda.synchronize()
    model.timed_op_ms()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model(dinp, "train", noise=dnoise)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    head_ms = model.timed_op_ms()
    elapsed = D.max_over_ranks(elapsed, device=dev)

    also = None
    if not args.no_extras:
        also = extras(args, model, hp, dinp, dnoise, inputs, rank, world, dev)

    if rank == 0:
        frames = world * hp.batch_size * hp.max_seq_len * args.steps
        value = frames / elapsed
        F = hp.batch_size * hp.n_nodes
        head_flops = 2.0 * hp.img_sz * hp.img_sz * hp.head_channels * hp.ngf * 9 * F      # algorithmic, per launch
        avg_ms = sum(head_ms) / len(head_ms)
        achieved = head_flops / (avg_ms * 1e-3) / 1e12
        line = {
            "metric": "predicted frames/sec, 64x64x3 seq_len=80 gcp_tree (inference forward, posterior path)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 25-room gcp_tree forward, 64x64x3, seq_len 80, batch 16/GPU, "
                                   "L=7 (127 nodes/seq decoded), discrete-logistic-mixture head, "
                                   + ("running-stat" if args.eval_bn else "batch-stat") + " BatchNorm",
                       "batch_per_gpu": hp.batch_size, "seq_len": hp.max_seq_len, "img": hp.img_sz,
                       "nodes_per_seq": hp.n_nodes, "parallelism": f"dp{world} (independent sequences, no collective)"},
            "roofline": {"kernel": "conv3x3_head_kernel<6, true> (decoder output head, 3x3 conv 16->100 ch @64x64 = 6 MFMA tiles + 4-channel 4x4x1 remainder, fused mixture mean)",
                         "bound": "mfma", "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                         # HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/r01_pmc_head_kernel.json):
                         # FETCH_SIZE 328209 KB x2 (gfx950 wide-load correction, MI355X_MICROARCH.md) + WRITE_SIZE 97536 KB;
                         # algorithmic bytes = 532.7 MB in (16 ch f32 @64x64 x 2032 frames) + 99.9 MB out
                         "traffic": HEAD_TRAFFIC_BYTES if (hp.batch_size == 16 and not args.eval_bn) else None,
                         "avg_launch_ms": round(avg_ms, 4), "flop_per_launch": head_flops},
        }
        if also is not None:
            line["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
