#!/usr/bin/env python
"""BASELINE.md section 2, in full: the CPU oracle timed on the host cores with threads = 1, 16 and all, configs c1 (B = 2) and c2
(B = 16), regions inference forward / planning rollout / training step, 3 warm-up + 10 timed iterations each, median and min
(about half an hour of box time).   python tools/cpu_baseline_full.py > gpurun_out/r05_cpu_baseline_full.json
(--reduced: the round-2 reduction — c2 on one thread at batch 2 with 1 + 3 iterations, c2 on all threads with 1 + 5 / 1 + 3.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    allt = torch.get_num_threads()
    regions = ("forward", "planning_rollout", "train_step")
    if "--reduced" in sys.argv:
        sched = [("c1", 2, k, r, 3, 10, 10, 1e9) for k in (1, allt) for r in regions]
        sched += [("c2", 16, allt, "forward", 1, 5, 5, 1e9), ("c2", 16, allt, "planning_rollout", 1, 5, 5, 1e9), ("c2", 16, allt, "train_step", 1, 3, 3, 1e9)]
        sched += [("c2", 2, 1, r, 1, 3, 3, 1e9) for r in regions]
        note = "BASELINE.md section 2 with the reductions named in tools/cpu_baseline_full.py's docstring (--reduced)"
    else:
        threads = sorted({1, min(16, allt), allt})
        sched = [(cfg, b, k, r, 3, 10, 10, 1e9) for cfg, b in (("c1", 2), ("c2", 16)) for k in threads for r in regions]
        note = "BASELINE.md section 2 in full: 3 warm-up + 10 timed iterations, threads 1 / 16 / all, c1 at B = 2 and c2 at B = 16"
    out = bench.cpu_baseline(schedule=sched)
    out["protocol"] = note + "; frames_per_s = batch * T / median seconds"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
