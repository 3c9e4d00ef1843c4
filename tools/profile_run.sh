#!/bin/bash
# kernel-trace of the headline forward and of the c2 training step (timelines only; no counters)
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
TAG=${1:-x}
rocprofv3 --kernel-trace --stats -d "$OUT/prof_fwd" -o p --output-format csv -- "$PY" bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > "$OUT/${TAG}_prof_fwd.json" 2>/dev/null
"$PY" tools/fwd_trace.py "$OUT/prof_fwd/p_kernel_trace.csv" 3 > "$OUT/${TAG}_fwd_trace.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/prof_train" -o p --output-format csv -- "$PY" tools/train_steps.py 4 > "$OUT/${TAG}_prof_train.txt" 2>/dev/null
"$PY" tools/trace_summary.py "$OUT/prof_train/p_kernel_trace.csv" 45 > "$OUT/${TAG}_train_trace.txt" 2>&1
rm -rf "$OUT"/prof_fwd/*.db "$OUT"/prof_train/*.db "$OUT"/prof_train/p_kernel_trace.csv
