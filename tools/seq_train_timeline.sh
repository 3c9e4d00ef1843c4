#!/bin/bash
# kernel-trace of the gcp_sequential training step (c2 shapes): per-queue timeline of one step (kernels >= MIN_US, default 20)
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
TAG=${1:-x}
MIN=${2:-20}
rocprofv3 --kernel-trace --stats -d "$OUT/prof_seq" -o p --output-format csv -- "$PY" tools/bench_sequential_train.py 3 > "$OUT/${TAG}_prof_seq.txt" 2>/dev/null
"$PY" tools/trace_summary.py "$OUT/prof_seq/p_kernel_trace.csv" 40 > "$OUT/${TAG}_seq_train_trace.txt" 2>&1
for q in 1 2 3 4; do
  echo "=== queue $q" >> "$OUT/${TAG}_seq_train_timeline.txt"
  "$PY" tools/trace_summary.py "$OUT/prof_seq/p_kernel_trace.csv" 0 $q $MIN | grep "t=" >> "$OUT/${TAG}_seq_train_timeline.txt"
done
rm -rf "$OUT"/prof_seq/*.db "$OUT"/prof_seq/p_kernel_trace.csv
