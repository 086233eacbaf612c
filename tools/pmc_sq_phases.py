#!/usr/bin/env python3
"""SQ counters of ONE train() under `rocprofv3 --pmc <SQ counters> --kernel-trace`, per kernel AND per training phase:
the dispatch sequence is cut into steps at every selection kernel (as tools/rocpd_phases.py does) and the counters of
one kernel are summed per range of steps.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave
(MI355X_MICROARCH.md); the ratios are what matters: wait = parked at s_waitcnt / barrier, wait_inst = issue stall,
active = issuing.
usage: tools/pmc_sq_phases.py run.db kernel_substring[,kernel_substring...] [edge edge ...]"""
import json
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2]
edges = [int(x) for x in sys.argv[3:]] or [0, 123, 261, 453, 746, 1301, 2483, 3529, 1 << 30]
SEL = ("k_select<", "k_rowsel_lean", "k_sel_lean", "k_chain_sel", "k_pool_sel")


def short(name):
    return name.split("(")[0].replace("void ", "").replace("bpe::", "").replace("bpe_g1::", "").replace("bpe_g4::", "")


cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
tcol = "start" if "start" in cols else None
q = f"select dispatch_id, name, counter_name, counter_value{', start, end' if tcol else ''} from pmc_events order by dispatch_id"
disp = {}
for row in cur.execute(q):
    d = disp.setdefault(row[0], {"name": short(row[1]), "c": {}, "dur": (row[5] - row[4]) / 1e3 if tcol else 0.0})
    d["c"][row[2]] = d["c"].get(row[2], 0.0) + float(row[3])
it = -1
acc = defaultdict(lambda: defaultdict(float))
for did in sorted(disp):
    d = disp[did]
    if d["name"].startswith(SEL):
        it += 1
    if not any(f in d["name"] for f in flt.split(",")) or it < 0:
        continue
    b = next(f"{lo}-{hi if hi < (1 << 30) else 'end'}" for lo, hi in zip(edges[:-1], edges[1:]) if lo <= it < hi)
    a = acc[b + " " + d["name"]]
    a["calls"] += 1
    a["dur_us_under_pmc"] += d["dur"]
    for k, v in d["c"].items():
        a[k] += v
out = {}
for b, a in acc.items():
    n = a["calls"]
    o = {"calls": int(n), "avg_us_under_pmc": round(a["dur_us_under_pmc"] / n, 1)}
    for k, v in a.items():
        if k not in ("calls", "dur_us_under_pmc"):
            o[k + "_per_call"] = round(v / n, 1)
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_FLAT", "SQ_WAIT_INST_LDS"):
            if k in a:
                o[k + "/WAVE_CYCLES"] = round(a[k] / wc, 3)
    if a.get("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
                  "SQ_INSTS_FLAT", "SQ_WAVE_CYCLES"):
            if k in a:
                o[k + "/wave"] = round(a[k] / a["SQ_WAVES"], 1)
    out[b] = o
print(json.dumps(out, indent=1))
