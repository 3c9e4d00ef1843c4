#!/bin/bash
# per-kernel time of bench.py's headline forward: tools/kstats.sh <tag> [bench args]  -> gpurun_out/<tag>_kernel_stats.csv + top list
TAG=$1; shift
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_ks" -o p --output-format csv -- "$PY" bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline "$@" > "$OUT/${TAG}_ks.log" 2>&1
cp "$OUT/${TAG}_ks"/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
rm -rf "$OUT/${TAG}_ks"/*.db
"$PY" - "$OUT/${TAG}_kernel_stats.csv" <<'PYEOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['TotalDurationNs'])/tot*100:5.1f}")
PYEOF
