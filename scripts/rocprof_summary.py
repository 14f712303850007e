#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db or kernel-trace CSV) per kernel:
calls, total / mean / median / min / max duration.  Used to produce profiles/*.md.

    python scripts/rocprof_summary.py <results.db | kernel_trace.csv> [--pmc]
"""
import csv
import sqlite3
import statistics
import sys
from collections import defaultdict


def rows_from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    for name, start, end, gx, gy, wx in cur.execute(
            "select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"):
        yield name, int(start), int(end), gx, gy, wx


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield (r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                   int(r.get("Grid_Size_X", 0) or 0), int(r.get("Grid_Size_Y", 0) or 0),
                   int(r.get("Workgroup_Size_X", 0) or 0))


def pmc_from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    try:
        return list(cur.execute("select name, counter_name, counter_value from pmc_events"))
    except Exception as e:      # schema differs between rocprofv3 builds
        return [("<pmc query failed: %s>" % e, "", 0)]


def short(name):
    name = name.replace("void ", "")
    for a, b in (("at::native::", ""), ("(anonymous namespace)::", "")):
        name = name.replace(a, b)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    path = sys.argv[1]
    rows = list(rows_from_db(path) if path.endswith(".db") else rows_from_csv(path))
    by = defaultdict(list)
    for name, s, e, gx, gy, wx in rows:
        by[(short(name), gx, gy, wx)].append(e - s)
    total = sum(sum(v) for v in by.values())
    span = rows[-1][2] - rows[0][1] if rows else 0
    print(f"kernels: {len(rows)}  sum of kernel time: {total / 1e6:.3f} ms  trace span: {span / 1e6:.3f} ms  "
          f"(GPU busy {100.0 * total / max(span, 1):.1f} %)\n")
    print("| kernel | grid (threads x rows) | block | calls | total ms | mean us | median us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for (name, gx, gy, wx), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{name}` | {gx} x {gy} | {wx} | {len(v)} | {sum(v) / 1e6:.3f} | {statistics.mean(v) / 1e3:.2f} | "
              f"{statistics.median(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {100.0 * sum(v) / total:.1f} |")
    if "--pmc" in sys.argv and path.endswith(".db"):
        agg = defaultdict(list)
        for name, counter, value in pmc_from_db(path):
            agg[(short(name), counter)].append(float(value))
        print("\n| kernel | counter | dispatches | mean per dispatch | sum |")
        print("|---|---|---|---|---|")
        for (name, counter), v in sorted(agg.items()):
            print(f"| `{name}` | {counter} | {len(v)} | {statistics.mean(v):.1f} | {sum(v):.1f} |")


if __name__ == "__main__":
    main()
