#!/bin/bash
# same-box A/B of the headline forward on this build and other builds of the library: bash tools/ab3.sh <rounds> <lib> [<lib> ...]
R=$1; shift
for i in $(seq $R); do
for lib in "" "$@"; do
  GCPX_LIB=$lib python tools/ab_bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-this build}', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
