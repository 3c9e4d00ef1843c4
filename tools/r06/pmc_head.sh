#!/bin/bash
# PMC passes over the standalone head timing (tools/r06/time_head32.py): per-kernel averages of the SQ counters -> gpurun_out/<tag>_pmc_head.txt
# usage (on the GPU box, from the repo root): bash tools/r06/pmc_head.sh <tag> [env assignments for the python program...]
set -u
TAG=$1; shift
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
PY=$(command -v python3)
mkdir -p "$OUT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAVES SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_head_$i -o p --output-format csv -- "$PY" tools/r06/time_head32.py > /tmp/pmc_head_$i.log 2>&1 || { echo "pmc pass $i failed" >&2; tail -3 /tmp/pmc_head_$i.log >&2; }
done
"$PY" - "$TAG" <<'PYEOF' > "$OUT/${TAG}_pmc_head.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/pmc_head_*/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            if "head" not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    print(name[:110])
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.0f}   ({len(v)} dispatches)")
PYEOF
cat "$OUT/${TAG}_pmc_head.txt" | head -80
