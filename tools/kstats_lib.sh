#!/bin/bash
# per-kernel average durations of the headline forward under rocprofv3 for this build and another one:
#   bash tools/kstats_lib.sh <other libgcpx.so> <kernel name substring> [...]
R=${GRAFT_REPO_ROOT:-$PWD}
OTHER=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in this other; do
  rm -rf /tmp/ks_$v
  if [ $v = other ]; then export GCPX_LIB=$OTHER; fi
  rocprofv3 --kernel-trace -d /tmp/ks_$v -o t --output-format csv -- python3 $R/bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline > /tmp/ks_$v.log 2>&1
  echo "== $v build"
  python3 $R/tools/kstats.py $(find /tmp/ks_$v -name "*kernel_trace.csv" | head -1) "$@"
done
