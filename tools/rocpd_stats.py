#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default ROCm 7.2 output) as the
per-kernel table `rocprofv3 --stats` prints: calls, total/avg/min/max ns, %.
usage: tools/rocpd_stats.py run_results.db [> profiles/xxx_kernel_stats.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(
    f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
    f"from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for name, calls, tot, avg, mn, mx in rows:
    print(f"\"{name}\",{calls},{tot},{avg:.1f},{mn},{mx},{100.0*tot/total:.2f}")
