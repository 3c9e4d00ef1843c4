#!/bin/bash
# kernel timeline of one training step per hardware queue (kernels / gaps above 120 us; the main queue also in full):
#   bash tools/train_lane.sh [tag]      (on the GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r02k}
export TMPDIR=/tmp
PY=$(command -v python3)
rocprofv3 --kernel-trace -d "$OUT/prof_train2" -o p --output-format csv -- "$PY" tools/train_steps.py 4 > /dev/null 2>&1
for q in 2 3 4 1; do
  echo "=== queue $q"
  "$PY" tools/trace_summary.py "$OUT/prof_train2/p_kernel_trace.csv" 0 $q 120 | tail -n +4 | cut -c1-110
done > "$OUT/${TAG}_train_lanes.txt" 2>&1
"$PY" tools/trace_summary.py "$OUT/prof_train2/p_kernel_trace.csv" 0 2 -1 | tail -n +4 | cut -c1-120 > "$OUT/${TAG}_train_main_lane.txt" 2>&1
rm -rf "$OUT/prof_train2"
cat "$OUT/${TAG}_train_lanes.txt"
