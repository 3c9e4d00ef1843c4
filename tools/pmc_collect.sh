#!/bin/bash
# rocprofv3 passes of bench.py's headline command on the GPU box (run from the repo root through gpurun):
#   1. --kernel-trace --stats            -> gpurun_out/prof_stats   (per-kernel average durations)
#   2. --pmc passes, one counter group each (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ: 8 slots)
# then tools/pmc_summary.py writes profiles/<round>_pmc_kernels.json and profiles/pmc_head_kernel.json.
# Counters are collected with --kernel-trace only (never with --sys-trace / hip / hsa domains), the program sits right after `--`.
set -u
ROUND=${1:-r03}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
CMD="bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats" -o p --output-format csv -- "$PY" $CMD > "$OUT/${ROUND}_prof_bench.json" 2> "$OUT/${ROUND}_prof_bench.err"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$i" -o p --output-format csv -- "$PY" $CMD > "$OUT/${ROUND}_pmc_$i.log" 2>&1 || echo "pmc pass $i failed" >&2
done
"$PY" tools/pmc_summary.py --round "$ROUND" --stats "$OUT"/prof_stats/*kernel_stats.csv --pmc "$OUT"/pmc_*/*counter_collection.csv
# only gpurun_out/ travels back from the GPU box: leave copies of what belongs under profiles/ there
cp profiles/${ROUND}_pmc_kernels.json profiles/pmc_head_kernel.json "$OUT"/ 2>/dev/null
cp "$OUT"/prof_stats/*kernel_stats.csv "$OUT/${ROUND}_kernel_stats.csv" 2>/dev/null
# gpurun_out/ is capped at 64 MiB: the raw counter tables (20 MB per SQ pass) stay on the GPU box, the summaries travel
rm -rf "$OUT"/pmc_[0-9]* "$OUT"/prof_stats
