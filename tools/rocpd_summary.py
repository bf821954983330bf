#!/usr/bin/env python3
"""Summaries of rocprofv3's rocpd SQLite output (ROCm 7.x default format).

    rocpd_summary.py stats <results.db>            -> CSV: kernel, calls, total_us, avg_us, pct
    rocpd_summary.py pmc <results.db> [...]        -> mean counter value per kernel per counter
"""
import collections
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    print("kernel,calls,total_us,avg_us,percent")
    for name, calls, total, avg, pct in c.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"):
        print(f"\"{name}\",{calls},{total:.3f},{avg:.3f},{pct:.3f}")


def pmc(dbs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for db in dbs:
        c = sqlite3.connect(db)
        for k, cn, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
            acc[k[:70]][cn].append(float(v))
    for k, d in acc.items():
        print(k)
        for cn, v in sorted(d.items()):
            print(f"    {cn:28s} n={len(v):4d} mean={sum(v) / len(v):18.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
