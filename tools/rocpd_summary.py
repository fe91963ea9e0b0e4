#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd/sqlite output) runs into small CSV files
for profiles/: per-kernel duration statistics (the `--kernel-trace --stats`
view) and per-kernel PMC counter averages (one `--pmc` pass per database).

  python tools/rocpd_summary.py stats  <results.db> > profiles/rNN_x_kernel_stats.csv
  python tools/rocpd_summary.py pmc    <results.db> [...more dbs] > profiles/rNN_x_pmc.csv
"""
import csv
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("idsp::", "")
    i = name.find("(")
    name = name[:i] if i > 0 else name
    return name.replace("void ", "")[:160]


def stats(db):
    con = sqlite3.connect(db)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_bytes"])
    rows = con.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
        "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    for r in rows:
        w.writerow([short(r[0]), r[1], f"{r[2]:.1f}", f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.2f}", f"{100 * r[2] / tot:.2f}", r[6], r[7], r[8]])


def pmc(dbs):
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "avg_value", "min_value", "max_value", "avg_dispatch_us"])
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute(
            "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(end - start)/1e3 "
            "from counters_collection group by kernel_name, counter_name order by 1, 2").fetchall()
        for r in rows:
            if "idsp" not in r[0]:
                continue
            w.writerow([short(r[0]), r[1], r[2], f"{r[3]:.4f}", f"{r[4]:.4f}", f"{r[5]:.4f}", f"{r[6]:.2f}"])


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
