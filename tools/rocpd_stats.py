#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max in ns) of a rocprofv3 rocpd database, as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/run_results.db > profiles/summary.csv
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    dur = "duration" if "duration" in cols else "(end - start)"
    rows = db.execute(f"select {name}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels "
                      f"group by {name} order by sum({dur}) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, t, a, lo, hi in rows:
        print(f'"{n}",{c},{t},{a:.1f},{100.0 * t / total:.2f},{lo},{hi}')


if __name__ == "__main__":
    main(sys.argv[1])
