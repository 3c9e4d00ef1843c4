#!/bin/bash
# PMC passes over the few-row LSTM GEMM (tools/r06/gemm_rows.py) -> gpurun_out/r06_gemm_rows_pmc.txt
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
mkdir -p "$OUT"
: > "$OUT/r06_gemm_rows_pmc.txt"
for rows in 16 64 256; do
 for mode in hot cold; do
  "$PY" tools/r06/gemm_rows.py $rows $mode >> "$OUT/r06_gemm_rows_pmc.txt"
  i=0
  rm -rf /tmp/pmc_gemm_*
  for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_gemm_$i -o p --output-format csv -- "$PY" tools/r06/gemm_rows.py $rows $mode > /tmp/pmc_gemm_$i.log 2>&1 || echo "pass $i failed: $(tail -2 /tmp/pmc_gemm_$i.log)" >> "$OUT/r06_gemm_rows_pmc.txt"
  done
  "$PY" - <<'PYEOF' >> "$OUT/r06_gemm_rows_pmc.txt"
import csv, glob, collections
acc = collections.defaultdict(list)
grid = set()
for path in glob.glob("/tmp/pmc_gemm_*/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if "gemm_kernel" not in (row.get("Kernel_Name") or ""):
                continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            grid.add((row.get("Grid_Size"), row.get("Workgroup_Size")))
print("   grid / workgroup:", sorted(grid))
for c, v in sorted(acc.items()):
    print(f"   {c:24s} {sum(v) / len(v):14.0f}   ({len(v)} dispatches)")
PYEOF
 done
done
cat "$OUT/r06_gemm_rows_pmc.txt"
