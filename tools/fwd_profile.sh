#!/bin/bash
# kernel trace of the headline forward only: tools/fwd_profile.sh <tag>  -> gpurun_out/<tag>_fwd_trace.txt (+ the bench line of the traced run)
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
TAG=${1:-x}
rocprofv3 --kernel-trace --stats -d "$OUT/prof_fwd_$TAG" -o p --output-format csv -- "$PY" bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > "$OUT/${TAG}_prof_fwd.json" 2>/dev/null
"$PY" tools/fwd_trace.py "$OUT/prof_fwd_$TAG/p_kernel_trace.csv" 3 > "$OUT/${TAG}_fwd_trace.txt" 2>&1
rm -rf "$OUT/prof_fwd_$TAG"
