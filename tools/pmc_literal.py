#!/usr/bin/env python3
"""The literal path (mode=0) under one rocprofv3 --pmc pass and one kernel trace of the same command
(tools/gpu_pmc_literal.sh): per kernel the HBM bytes (32 B x the size-weighted TCC/EA request counters, tools/pmc_summary.py),
the average duration of the kernel trace, and the physical fraction of the 8 TB/s peak.
usage: tools/pmc_literal.py pmc.db kernel_stats.csv merges out.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_summary as ps
import bench
db, stats, merges, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
t, d = ps.table(db), ps.durations(stats)
rows, tot_b, tot_us = {}, 0.0, 0.0
for k, e in t.items():
    if k not in d:
        continue
    calls, us = d[k]
    b = e[ps.RD] + e[ps.WR] + e[ps.AT]
    if calls < merges // 2:  # (the loop's own kernels run once per iteration)
        continue
    rows[k] = {"calls": calls, "avg_us": round(us / calls, 2), "read_bytes_per_call": int(e[ps.RD] / e["calls"]),
               "write_bytes_per_call": int(e[ps.WR] / e["calls"]), "atomics_per_call": int(e[ps.AT] / 32 / e["calls"]),
               "TBps": round(b / e["calls"] / (us / calls) / 1e6, 3), "frac_of_hbm_peak": round(b / e["calls"] / (us / calls) / 1e6 / 8.0, 3)}
    tot_b += b / e["calls"]
    tot_us += us / calls
out = {"workload": f"regex1g, mode=0 (the reference's loop as written), first {merges} merges", "source_hash": bench.source_hash(),
       "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"])),
       "iteration": {"hbm_bytes": int(tot_b), "us": round(tot_us, 1), "TBps": round(tot_b / tot_us / 1e6, 3), "physical_frac_of_hbm_peak": round(tot_b / tot_us / 1e6 / 8.0, 3)}}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out["iteration"]), {k: (v["avg_us"], v["frac_of_hbm_peak"], v["atomics_per_call"]) for k, v in out["kernels"].items()})
