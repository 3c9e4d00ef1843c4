#!/usr/bin/env python
"""Per-kernel rocprofv3 summary of bench.py's headline command: average duration (--stats pass) and PMC counters (--pmc passes)
for the kernels the roofline discussion names, written to profiles/<round>_pmc_kernels.json; the head kernel's HBM traffic
(FETCH_SIZE x 2 for the gfx950 wide-load undercount + WRITE_SIZE, /opt/skills/guides/MI355X_MICROARCH.md "HBM") also goes to
profiles/pmc_head_kernel.json together with a hash of the kernel sources, which is what bench.py prints as roofline.traffic.

usage (tools/pmc_collect.sh calls it): python tools/pmc_summary.py --round r02 --stats <kernel_stats.csv> --pmc <counter_collection.csv> ...
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = {"head": "conv3x3_head", "up16": "conv3x3_up16", "up32": "conv3x3_up32", "conv3x3_tiled": "conv3x3_kernel<", "gemm": "gemm_kernel<",
           "gemm_split": "gemm_split_kernel", "gemm_planes": "gemm_planes_kernel", "split_rows": "split_rows_kernel", "level_pre": "level_pre_kernel",
           "mlp": "mlp_kernel<", "mlp_group": "mlp_group_kernel", "enc_lds": "conv4x4s2_lds_kernel",
           "enc_split": "conv4x4s2_split_kernel", "enc_image": "conv4x4s2_image_kernel"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("--stats", nargs="*", default=[])
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    out = {k: {"match": pat, "variants": {}} for k, pat in KERNELS.items()}
    for path in a.stats:
        with open(path) as f:
            for row in csv.DictReader(f):
                for k, pat in KERNELS.items():
                    if pat in row["Name"]:
                        v = out[k]["variants"].setdefault(row["Name"], {})
                        v.update(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3, total_us=float(row["TotalDurationNs"]) / 1e3)
    for path in a.pmc:
        acc = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                for k, pat in KERNELS.items():
                    if pat in row["Kernel_Name"]:
                        acc.setdefault((k, row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        for (k, name, ctr), vals in acc.items():
            v = out[k]["variants"].setdefault(name, {})
            v.setdefault("pmc_mean_per_dispatch", {})[ctr] = sum(vals) / len(vals)
            v["pmc_dispatches"] = len(vals)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", f"{a.round}_pmc_kernels.json"), "w") as f:
        json.dump(out, f, indent=1)
    head = [v for v in out["head"]["variants"].values() if "pmc_mean_per_dispatch" in v]
    if head:
        p = head[0]["pmc_mean_per_dispatch"]
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            import bench
            d = {"kernel_src_sha": bench.kernel_source_sha(), "sources": list(bench.HEAD_KERNEL_SOURCES), "batch": a.batch, "eval_bn": False, "with_loss": True,
                 "FETCH_SIZE_KB": p["FETCH_SIZE"], "WRITE_SIZE_KB": p["WRITE_SIZE"],
                 "traffic_bytes_per_launch": int((2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024),
                 "rule": "FETCH_SIZE x 2 (gfx950 counts a wide coalesced read at half its bytes) + WRITE_SIZE, KB -> bytes",
                 "avg_us_under_profiler": head[0].get("avg_us"), "round": a.round, "counters": p}
            with open(os.path.join(ROOT, "profiles", "pmc_head_kernel.json"), "w") as f:
                json.dump(d, f, indent=1)
            print("head kernel traffic per launch:", d["traffic_bytes_per_launch"] / 1e6, "MB")
    print("wrote profiles/%s_pmc_kernels.json" % a.round)


if __name__ == "__main__":
    main()
