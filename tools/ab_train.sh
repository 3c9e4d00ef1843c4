#!/bin/bash
# same-box A/B of the training step (tools/bench_train.py, first two lines) with / without an environment setting:
#   bash tools/ab_train.sh VAR=value [repeats]
R=${2:-3}
for i in $(seq $R); do
  TOP=0 python tools/bench_train.py 2>/dev/null | grep "training step" | sed 's/^/base       /'
  env $1 TOP=0 python tools/bench_train.py 2>/dev/null | grep "training step" | sed "s/^/$1 /"
done
