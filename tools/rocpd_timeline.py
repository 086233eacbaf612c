#!/usr/bin/env python3
"""Print the kernel dispatches of a rocprofv3 rocpd database in time order (name, start offset,
duration, gap to the previous kernel's end), optionally only the last N.
usage: tools/rocpd_timeline.py run_results.db [last_n] [min_duration_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
if last_n:
    rows = rows[-last_n:]
t0 = rows[0][1]
prev_end = t0
for name, s, e in rows:
    short = name.split("(")[0].replace("void ", "").replace("bpe::", "")
    if (e - s) / 1e3 >= min_us or (s - prev_end) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:12.1f} us  dur {(e - s) / 1e3:10.1f}  gap {(s - prev_end) / 1e3:9.1f}  {short}")
    prev_end = e
