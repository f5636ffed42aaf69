#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the same columns `--stats` prints:
name, calls, total ns, average ns, percentage (+ vgpr/lds of the dispatches).
usage: tools/rocpd_summary.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage",
                "VGPRs", "AccumVGPRs", "SGPRs", "LDSBytes", "MaxGridX", "WorkgroupX"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / total, 3), *r[6:]])


if __name__ == "__main__":
    main()
