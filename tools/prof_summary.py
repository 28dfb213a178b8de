#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs (kernel-trace --stats and --pmc passes) into small text files
for profiles/.   usage: prof_summary.py <results.db> [<results.db> ...] > summary.txt"""
import sqlite3
import sys


def short(name):
    for k in ("k_qp3f", "k_qp3", "k_qp2", "k_qp4", "k_qp5", "k_qp", "k_step_m", "k_step", "k_init_m", "k_init", "k_warm_jerk", "k_sample", "k_rnea_batch", "k_eval_constraints"):
        if k in name:
            i = name.find("ILi")
            return k + ("<%s>" % name[i + 3:name.find("E", i)] if i >= 0 else "")
    return name[:60]


for path in (sys.argv[1:] if __name__ == "__main__" else []):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("== %s" % path)
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                       "max(workgroup_x), max(grid_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-28s %6s %12s %12s %12s %12s %6s %5s %5s %5s %7s %8s %5s %8s" % (
        "kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds_B", "scratch", "wg", "grid"))
    for r in rows[:12]:
        print("%-28s %6d %12.3f %12.1f %12.1f %12.1f %6.2f %5s %5s %5s %7s %8s %5s %8s" % (
            short(r[0]), r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    try:
        crow = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                           "group by kernel_name, counter_name order by sum(value) desc").fetchall()
    except sqlite3.Error:
        crow = []
    if crow:
        print("-- counters (per-dispatch average, raw units as reported by rocprofv3)")
        for r in crow[:40]:
            print("%-28s %-14s dispatches=%-5d avg=%-16.1f sum=%.1f" % (short(r[0]), r[1], r[2], r[3], r[4]))
    print()
