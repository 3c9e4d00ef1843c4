#!/bin/bash
# same-box A/B of the headline forward with / without one environment setting: bash tools/ab_env.sh VAR=value [repeats]
R=${2:-3}
for i in $(seq $R); do
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base      ', d['ms_per_step'])"
  env $1 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"
done
