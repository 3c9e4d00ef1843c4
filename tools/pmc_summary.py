#!/usr/bin/env python
"""Average rocprofv3 PMC counters per kernel from counter_collection CSVs:
   python tools/pmc_summary.py <kernel-name-substring> gpurun_out/pmc_*/**/*counter_collection.csv > profiles/rNN_pmc_<kernel>.json"""
import csv
import glob
import json
import sys


def main():
    pat = sys.argv[1]
    out = {}
    for arg in sys.argv[2:]:
        for path in glob.glob(arg, recursive=True):
            vals, disp = {}, set()
            with open(path) as f:
                for row in csv.DictReader(f):
                    if pat not in row["Kernel_Name"]:
                        continue
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    disp.add(row["Dispatch_Id"])
            for k, v in vals.items():
                out[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(disp), "file": path.split("/")[-1]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
