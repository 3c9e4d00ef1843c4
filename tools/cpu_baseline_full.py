#!/usr/bin/env python
"""the CPU oracle timed on the host cores with threads = 1 and all, configs c1 and c2, regions inference
forward / planning rollout / training step, median and min.  Deviation (stated in the output): c2 with ONE thread runs at batch 2
with 1 warm-up + 3 timed iterations, and c2 on all threads with 1 warm-up + 5 (3 for the training step) — the protocol's
3 + 10 at B=16 on one thread is hours of box time.   python tools/cpu_baseline_full.py > gpurun_out/r02_cpu_baseline_full.json"""
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
    sched = [("c1", 2, k, r, 3, 10, 10, 1e9) for k in (1, allt) for r in ("forward", "planning_rollout", "train_step")]
    sched += [("c2", 16, allt, "forward", 1, 5, 5, 1e9), ("c2", 16, allt, "planning_rollout", 1, 5, 5, 1e9), ("c2", 16, allt, "train_step", 1, 3, 3, 1e9)]
    sched += [("c2", 2, 1, r, 1, 3, 3, 1e9) for r in ("forward", "planning_rollout", "train_step")]
    out = bench.cpu_baseline(schedule=sched)
    out["protocol"] = ("BASELINE.md section 2 with the reductions named in tools/cpu_baseline_full.py's docstring; frames_per_s = batch * T / "
                       "median seconds")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
