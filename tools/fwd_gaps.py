"""Where one forward's time goes on the critical queue: python tools/fwd_gaps.py <kernel_trace.csv>
(busy time, sum and histogram of the gaps between consecutive kernels of the queue that carries the output head)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = 'loss_combine' if any('loss_combine' in r['Kernel_Name'] for r in rows) else 'conv3x3_head'
ends = [i for i, r in enumerate(rows) if last in r['Kernel_Name']]
a, b = ends[-2], ends[-1]
seg = rows[a + 1:b + 1]
q = collections.Counter(r['Queue_Id'] for r in seg if 'conv3x3_head' in r['Kernel_Name']).most_common(1)[0][0]
mine = [r for r in seg if r['Queue_Id'] == q]
t0 = int(rows[a]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in mine) / 1e3
gaps = []
prev = t0
for r in mine:
    gaps.append((int(r['Start_Timestamp']) - prev) / 1e3)
    prev = int(r['End_Timestamp'])
span = (int(seg[-1]['End_Timestamp']) - t0) / 1e3
print(f"forward span {span:.1f} us; queue {q}: {len(mine)} kernels, busy {busy:.1f} us, gaps {sum(gaps):.1f} us")
h = collections.Counter(min(int(g // 1), 20) for g in gaps)
print("gap histogram (us: count):", {k: h[k] for k in sorted(h)})
big = sorted(((g, r['Kernel_Name'][:60]) for g, r in zip(gaps, mine)), reverse=True)[:10]
for g, n in big:
    print(f"  {g:7.1f} us before {n}")
