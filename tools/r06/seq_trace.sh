cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/seqtrace; mkdir -p /root/repo/gpurun_out/seqtrace
cd /root/repo
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seqtrace -- python3 tools/bench_sequential_train.py 4 > gpurun_out/seqtrace/log.txt 2>&1
f=$(find gpurun_out/seqtrace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $f 12 > gpurun_out/seqtrace/summary.txt 2>&1
for q in 1 2 3 4; do echo "== queue $q" >> gpurun_out/seqtrace/summary.txt; python3 tools/trace_summary.py $f 0 $q 60 | tail -n +4 | head -150 >> gpurun_out/seqtrace/summary.txt; done
python3 - $f <<'PY' >> gpurun_out/seqtrace/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'radam_tick_kernel' in r['Kernel_Name']]
seg = rows[idx[-2]:idx[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
# coarse timeline: per 0.5 ms bucket, busy time per queue
import collections
B = 500000
buckets = collections.defaultdict(lambda: collections.defaultdict(int))
for r in seg:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    b = s // B
    while b * B < e:
        lo, hi = max(s, b * B), min(e, (b + 1) * B)
        buckets[b][r['Queue_Id']] += hi - lo
        b += 1
print("== busy fraction per 0.5 ms bucket and queue")
qs = sorted({r['Queue_Id'] for r in seg})
for b in sorted(buckets):
    print("%5.1f ms " % (b * 0.5) + "  ".join("q%s %3d%%" % (q, 100 * buckets[b][q] // B) for q in qs))
PY
rm -f $f gpurun_out/seqtrace/*/*.csv 2>/dev/null; find gpurun_out/seqtrace -name "*.csv" -delete
