#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time stats and PMC counter sums.
usage: rocpd_summary.py <results.db> [more.db ...]   (prints markdown)"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"### {path}\n")
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | lds B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {short(r[0])} | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | "
              f"{100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} |")
    try:
        pm = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                         "group by kernel_name, counter_name order by sum(value) desc").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print("\n| kernel | counter | dispatches | sum | avg per dispatch |")
        print("|---|---|---|---|---|")
        for r in pm:
            print(f"| {short(r[0])} | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]:.1f} |")
    print()
