#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd .db or *_kernel_stats.csv) into a short markdown table.

    python tools/rocprof_summary.py gpurun_out/prof_r1 > profiles/r01_kernel_stats.md
"""
import csv
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*?>)?)\(", name)
    if m and len(m.group(1)) < 90:
        return m.group(1)
    return name[:80] + ("..." if len(name) > 80 else "")


def rows_from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    return [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4]))
            for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                        float(r["Percentage"])))
    return out


def main():
    d = sys.argv[1]
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    csvs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = rows_from_csv(csvs[0]) if csvs else rows_from_db(dbs[0])
    rows.sort(key=lambda r: -r[2])
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in rows:
        if pct < 0.05:
            continue
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (short(name), calls, total, avg, pct))
    if dbs and not csvs:
        by_grid(dbs[0])


def by_grid(path):
    """The same kernel name is launched with different geometries (one utterance in the timed steps, 8 in the
    beyond-cache roofline leg): average per (kernel, grid) for the kernels that have more than one."""
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, grid_x, grid_y, grid_z, count(*), avg(end - start) from kernels "
                            "group by name, grid_x, grid_y, grid_z"))
    names = {}
    for r in rows:
        names.setdefault(r[0], []).append(r)
    multi = {k: v for k, v in names.items() if len(v) > 1 and ("assx" in k or "_kernel" in k) and "at::" not in k}
    if not multi:
        return
    print()
    print("| kernel | grid | calls | avg us |")
    print("|---|---|---:|---:|")
    for k, v in sorted(multi.items(), key=lambda kv: -sum(r[4] * r[5] for r in kv[1])):
        for r in sorted(v, key=lambda r: -r[4]):
            print("| `%s` | %d x %d x %d | %d | %.2f |" % (short(k), r[1], r[2], r[3], r[4], r[5] / 1e3))


if __name__ == "__main__":
    main()
