#!/bin/bash
# same-box A/B of the headline forward on two builds of the library: bash tools/ab_lib.sh <other libgcpx.so> [repeats]
R=${2:-3}
for i in $(seq $R); do
  python tools/ab_bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this build ', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  GCPX_LIB=$1 python tools/ab_bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('other build', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
