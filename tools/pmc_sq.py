#!/usr/bin/env python3
"""per-kernel means of the counters of one rocprofv3 --pmc pass.  usage: pmc_sq.py run.db [name filter]"""
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = cur.execute("select name, counter_name, counter_value from pmc_events").fetchall()
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name, cn, v in rows:
    key = name.split("(")[0].replace("void ", "").replace("bpe::", "")
    if flt and flt not in key:
        continue
    a = acc[key][cn]
    a[0] += 1
    a[1] += float(v)
for k, d in acc.items():
    print(k, {cn: (n, round(s / n, 1)) for cn, (n, s) in sorted(d.items())})
