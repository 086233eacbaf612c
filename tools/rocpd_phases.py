#!/usr/bin/env python3
"""Per-kernel device time by training phase from a rocprofv3 rocpd database of ONE train() call:
the dispatch sequence is cut into iterations at every selection kernel (k_select / k_rowsel_lean /
k_sel_lean) and summed per range of iterations.
usage: tools/rocpd_phases.py run_results.db [edge edge ...]  > profiles/xxx_phases.json"""
import json
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
edges = [int(x) for x in sys.argv[2:]] or [0, 10, 100, 300, 1000, 2000, 4000, 8000, 16000, 24000, 1 << 30]


def short(name):
    s = name.split("(")[0].replace("void ", "").replace("bpe::", "").replace("bpe_g1::", "").replace("bpe_g4::", "")
    return s


SEL = ("k_select<", "k_rowsel_lean", "k_sel_lean", "k_chain_sel", "k_pool_sel")
it = -1
per = defaultdict(lambda: defaultdict(lambda: [0, 0.0, []]))  # bin -> kernel -> [calls, us, durations]
span = defaultdict(lambda: [None, None, 0])
for name, s, e in rows:
    k = short(name)
    if k.startswith(SEL):
        it += 1
    if it < 0:
        b = "setup"
    else:
        b = next(f"{lo}-{hi if hi < (1 << 30) else 'end'}" for lo, hi in zip(edges[:-1], edges[1:]) if lo <= it < hi)
    c = per[b][k]
    c[0] += 1
    c[1] += (e - s) / 1e3
    c[2].append((e - s) / 1e3)
    sp = span[b]
    sp[0] = s if sp[0] is None else sp[0]
    sp[1] = e
    sp[2] = max(sp[2], it + 1)
out = {}
prev_it = 0
for b in per:
    ks = sorted(per[b].items(), key=lambda kv: -kv[1][1])
    n_it = span[b][2] - prev_it if b != "setup" else 0
    if b != "setup":
        prev_it = span[b][2]
    out[b] = {"iterations": n_it, "wall_us": round((span[b][1] - span[b][0]) / 1e3, 1),
              "us_per_iteration": round((span[b][1] - span[b][0]) / 1e3 / max(n_it, 1), 2),
              "kernels": {k: {"calls": v[0], "total_us": round(v[1], 1), "avg_us": round(v[1] / v[0], 2),
                                  "p10_p50_p90_us": [round(sorted(v[2])[int(q * (len(v[2]) - 1))], 2) for q in (0.1, 0.5, 0.9)]}
                              for k, v in ks}}
print(json.dumps(out, indent=1))
