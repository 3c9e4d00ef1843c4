#!/bin/bash
# same-box A/B of the c2 training step (tools/bench_train.py's first two lines) on this build and other builds: bash tools/ab_train_lib.sh <rounds> <lib> ...
R=$1; shift
for i in $(seq $R); do
for lib in "" "$@"; do
  echo "${lib:-this build}: $(GCPX_LIB=$lib python tools/bench_train.py c2 2>/dev/null | head -2 | tr '\n' ' ')"
done; done
