#!/usr/bin/env python3
"""Turns a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite, ROCm 7.2 default output) into the
plain-text per-kernel summary that is committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_x/NAME_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start)/1e6, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[5] for r in rows) or 1.0
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-58s %6s %10s %10s %10s %10s %6s %5s %5s %6s %7s %5s %9s %4s" % (
        "kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "pct", "vgpr", "agpr", "sgpr", "lds", "scr", "grid_x", "wg"))
    for r in rows:
        name = r[0].split("(")[0][:58]
        print("%-58s %6d %10.1f %10.1f %10.1f %10.3f %6.1f %5s %5s %6s %7s %5s %9s %4s" % (
            (name,) + r[1:6] + (100.0 * r[5] / total,) + r[6:]))


if __name__ == "__main__":
    main(sys.argv[1])
