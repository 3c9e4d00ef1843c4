"""Summarise one training step from a rocprofv3 kernel trace CSV: python tools/trace_summary.py <kernel_trace.csv> [top]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'radam_tick_kernel' in r['Kernel_Name']]      # one per optimizer step (its last launch but the re-pack)
seg = rows[idx[-2]:idx[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
print("step wall (tick to tick): %.2f ms; kernels: %d" % ((int(seg[-1]['End_Timestamp']) - t0) / 1e6, len(seg)))
q = collections.defaultdict(int)
k = collections.defaultdict(lambda: [0, 0])
for r in seg:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    q[r['Queue_Id']] += d
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    k[nm][0] += d; k[nm][1] += 1
print("busy ms per queue:", {a: round(b / 1e6, 2) for a, b in q.items()}, "sum %.1f" % (sum(q.values()) / 1e6))
# union of busy intervals (any queue)
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("GPU non-idle %.2f ms" % (busy / 1e6))
for nm, (d, n) in sorted(k.items(), key=lambda kv: -kv[1][0])[:top]:
    print("  %-72s %4d calls %8.2f ms" % (nm, n, d / 1e6))

if len(sys.argv) > 3:      # timeline of one queue: python tools/trace_summary.py trace.csv 0 <queue id> [min_us]
    qid, min_us = sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 100.0
    prev_end = t0
    for r in seg:
        if r['Queue_Id'] != qid:
            continue
        s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap, dur = (s_ - prev_end) / 1e3, (e_ - s_) / 1e3
        if gap > min_us or dur > min_us:
            print("  t=%8.1f us  gap %7.1f  dur %7.1f  %s" % ((s_ - t0) / 1e3, gap, dur, r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]))
        prev_end = max(prev_end, e_)
