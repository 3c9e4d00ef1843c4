#!/bin/bash
# kernel-trace of the c2 training step with the per-queue timeline of one step (every kernel >= MIN_US, default 8)
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
TAG=${1:-x}
MIN=${2:-8}
rocprofv3 --kernel-trace --stats -d "$OUT/prof_train" -o p --output-format csv -- "$PY" tools/train_steps.py 4 > "$OUT/${TAG}_prof_train.txt" 2>/dev/null
"$PY" tools/trace_summary.py "$OUT/prof_train/p_kernel_trace.csv" 45 > "$OUT/${TAG}_train_trace.txt" 2>&1
for q in 1 2 3 4; do
  echo "=== queue $q" >> "$OUT/${TAG}_train_timeline.txt"
  "$PY" tools/trace_summary.py "$OUT/prof_train/p_kernel_trace.csv" 0 $q $MIN | grep "t=" >> "$OUT/${TAG}_train_timeline.txt"
done
rm -rf "$OUT"/prof_train/*.db "$OUT"/prof_train/p_kernel_trace.csv
