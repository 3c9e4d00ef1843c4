"""Timeline of one forward from a rocprofv3 kernel trace of bench.py: python tools/fwd_trace.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one forward = from the end of the previous forward's last kernel (the loss combination when the forward computes its losses, else
# the output head) to the end of the last one
last = 'loss_combine' if any('loss_combine' in r['Kernel_Name'] for r in rows) else 'conv3x3_head'
heads = [i for i, r in enumerate(rows) if last in r['Kernel_Name']]
a, b = heads[-2], heads[-1]
seg = rows[a + 1:b + 1]
t0 = int(rows[a]['End_Timestamp'])
print("forward span %.1f us, %d kernels" % ((int(seg[-1]['End_Timestamp']) - t0) / 1e3, len(seg)))
prev = t0
for r in seg:
    s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (e_ - s_) / 1e3 >= min_us or (s_ - prev) / 1e3 >= min_us:
        print("  t=%7.1f  q%s  idle-before %6.1f  dur %7.1f  %s" % ((s_ - t0) / 1e3, r['Queue_Id'], max(0, (s_ - prev) / 1e3), (e_ - s_) / 1e3,
              r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:58]))
    prev = max(prev, e_)
