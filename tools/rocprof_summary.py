#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into a small CSV: name, calls, total ms, avg us, %.

    python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.csv
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ms,avg_us,min_us,max_us,percent")
    for name, n, t, a, lo, hi in rows:
        print(f"\"{short(name)}\",{n},{t/1e6:.3f},{a/1e3:.2f},{lo/1e3:.2f},{hi/1e3:.2f},{100*t/tot:.2f}")
    # the GEMMs split by grid size (M = 2438 launches vs the batched M = 4876 launches)
    rows = con.execute("select name, grid_x, count(*), avg(end-start) from kernels where name like '%gemm_bt_kernel%' "
                       "group by name, grid_x order by name, grid_x").fetchall()
    print("\nkernel,grid_x,calls,avg_us")
    for name, gx, n, a in rows:
        print(f"\"{short(name)}\",{gx},{n},{a/1e3:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
