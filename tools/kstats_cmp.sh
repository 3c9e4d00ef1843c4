#!/bin/bash
# per-kernel durations of the c2 training step (forward kernels included) under rocprofv3 for this build and another one, side by side,
# largest change in time per step first:   bash tools/kstats_cmp.sh <other libgcpx.so> [script args...]
R=${GRAFT_REPO_ROOT:-$PWD}
OTHER=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in this other; do
  rm -rf /tmp/kc_$v
  if [ $v = other ]; then case $OTHER in /*) export GCPX_LIB=$OTHER;; *) export GCPX_LIB=$R/$OTHER;; esac; fi
  rocprofv3 --kernel-trace -d /tmp/kc_$v -o t --output-format csv -- python3 $R/tools/train_steps.py "$@" > /tmp/kc_$v.log 2>&1 || tail -5 /tmp/kc_$v.log
done
python3 - <<'PY'
import csv, glob, collections
def load(d):
    f = glob.glob(f"/tmp/kc_{d}/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:100]
        acc[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return acc
a, b = load("this"), load("other")
rows = []
for n in set(a) | set(b):
    ta, tb = sum(a.get(n, [])), sum(b.get(n, []))
    rows.append((ta - tb, ta, tb, len(a.get(n, [])), n))
rows.sort(key=lambda r: -abs(r[0]))
tot_a, tot_b = sum(r[1] for r in rows), sum(r[2] for r in rows)
print(f"total kernel time: this {tot_a / 1e3:.2f} ms, other {tot_b / 1e3:.2f} ms (whole run)")
for d, ta, tb, c, n in rows[:40]:
    print(f"{d / 1e3:+8.3f} ms  this {ta / 1e3:8.3f}  other {tb / 1e3:8.3f}  x{c:5d}  {n}")
PY
