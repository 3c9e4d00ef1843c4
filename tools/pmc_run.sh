#!/bin/bash
# rocprofv3 counter passes of an arbitrary python command on the GPU box: tools/pmc_run.sh <tag> <script.py> [args...]
# -> gpurun_out/<tag>_pmc.txt (per kernel: mean of every counter per dispatch) and gpurun_out/<tag>_stats.csv
set -u
TAG=$1; shift
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o p --output-format csv -- "$PY" "$@" > "$OUT/${TAG}_run.log" 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp -d "$OUT/${TAG}_pmc_$i" -o p --output-format csv -- "$PY" "$@" > "$OUT/${TAG}_pmc_$i.log" 2>&1 || echo "pmc pass $i failed" >&2
done
"$PY" - "$OUT" "$TAG" <<'PYEOF'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/{tag}_pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/{tag}_pmc.txt", "w") as fh:
    for k, cs in acc.items():
        if "gcpx" not in k and "conv" not in k and "gemm" not in k and "mlp" not in k:
            continue
        fh.write(k[:140] + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f} max={max(v):16.1f}\n")
print(open(f"{out}/{tag}_pmc.txt").read())
PYEOF
cp "$OUT/${TAG}_stats"/*kernel_stats.csv "$OUT/${TAG}_stats.csv" 2>/dev/null
# gpurun_out/ is capped at 64 MiB: the raw counter tables stay on the GPU box, the summaries travel
rm -rf "$OUT"/${TAG}_pmc_[0-9]* "$OUT"/${TAG}_stats
