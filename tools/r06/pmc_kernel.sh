#!/bin/bash
# one PMC pass over bench.py's forward for the kernels matching a pattern: bash tools/r06/pmc_kernel.sh <pattern> [ENV=...]
set -u
PAT=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
PY=$(command -v python3)
rm -rf /tmp/pmc_k
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_k -o p --output-format csv -- "$PY" bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
"$PY" - "$PAT" <<'PYEOF'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/pmc_k/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or ""
            if sys.argv[1] in name:
                acc[name[:90] + " grid=" + str(row.get("Grid_Size"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} {sum(v) / len(v):16.0f}   ({len(v)})")
PYEOF
