"""Per-kernel average durations from a rocprofv3 --kernel-trace CSV: python tools/kstats.py <kernel_trace.csv> [substring ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
acc = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if pats and not any(p in n for p in pats):
        continue
    key = (n[:90], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')))
    acc.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, g, w), v in acc.items():
    v2 = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, len(v) - len(v) // 4)]
    print(f"{sum(v2) / len(v2):9.1f} us  x{len(v):4d}  grid {g:>9s} wg {w:>4s}  {n}")
