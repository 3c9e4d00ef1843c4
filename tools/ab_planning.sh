#!/bin/bash
# same-box A/B of one CEM iteration on latents (tools/bench_planning.py) with / without an environment setting:
#   bash tools/ab_planning.sh VAR=value [repeats]
R=${2:-3}
for i in $(seq $R); do
  python tools/bench_planning.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base      ', d['ms_per_iteration'])"
  env $1 python tools/bench_planning.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_iteration'])"
done
