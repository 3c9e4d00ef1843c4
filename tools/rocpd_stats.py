#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) as a per-kernel stats table (like --stats CSV).

usage: python tools/rocpd_stats.py gpurun_out/prof/xxx_results.db [--md] > profiles/rNN_kernel_stats.md
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("""
        select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
               max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# kernel stats from {path.split('/')[-1]} (durations in us; total GPU kernel time {total / 1e3:.1f} us)\n")
    print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | agpr | lds B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx, vg, ag, lds, grid, wg in rows:
        nm = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{nm}` | {n} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.1f} | {vg} | {ag} | {lds} | {grid} | {wg} |")


if __name__ == "__main__":
    main()
