"""Forward-to-forward spans from a rocprofv3 kernel trace (head end to head end) and what fills the hand-over between two
forwards: python tools/fwd_span_stats.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
heads = [i for i, r in enumerate(rows) if 'conv3x3_head' in r['Kernel_Name']]
spans, tails = [], []
for a, b in zip(heads[10:-1], heads[11:]):
    t0 = int(rows[a]['End_Timestamp'])
    spans.append((int(rows[b]['End_Timestamp']) - t0) / 1e3)
    # hand-over: from the head's end to the start of the first encoder conv of the next forward
    seg = rows[a + 1:b]
    first_conv = next(r for r in seg if 'conv4x4s2' in r['Kernel_Name'])
    tails.append(((int(first_conv['Start_Timestamp']) - t0) / 1e3, [(r['Kernel_Name'].replace('void ', '')[:40], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in seg if int(r['Start_Timestamp']) <= int(first_conv['Start_Timestamp'])]))
print("forwards %d: span median %.1f us, min %.1f, max %.1f" % (len(spans), st.median(spans), min(spans), max(spans)))
print("head end -> first encoder conv: median %.1f us" % st.median(t for t, _ in tails))
for nm, s_, d in tails[len(tails) // 2][1]:
    print("   t=%7.1f dur %6.1f  %s" % (s_, d, nm))
