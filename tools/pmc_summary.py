#!/usr/bin/env python3
"""Reduce ONE rocprofv3 --pmc pass of a bench command to the HBM bytes its kernels move.

Counters (gfx950, raw, size-weighted: a 64-byte request counts 2, a 128-byte one 4):
    TCC_EA0_RDREQ_DRAM_32B  TCC_EA0_WRREQ_WRITE_DRAM_32B  TCC_EA0_WRREQ_WRITE_ATOMIC_32B
bytes = 32 x count.  Calibrated on kernels with known byte counts in this engine's access patterns
(tools/pmc_calib.hip, profiles/r4_pmc_calibration.json): reads 1.00 x requested for wide and narrow coalesced streams,
1.05 x for the merge pass's slot + header + mask pattern; every read request is 128 bytes (a random 32-byte record
costs 128, a 4-byte gather 128); writes exact; a device atomic counts as one 32-byte write.  The derived FETCH_SIZE
tallies a 128-byte request at 64 (x2 for EVERY read pattern, not only wide streams), WRITE_SIZE = writes + atomics.

usage: tools/pmc_summary.py raw.db workload_name out.json [n_bytes_of_the_input [merges_per_train [kernel_stats.csv [atomic_peak.json]]]]

With the kernel-trace summary of the SAME command (tools/rocpd_stats.py, its own rocprofv3 run: durations under a PMC
pass are not the kernel's) every kernel gets its own line -- bytes, atomics, average duration, fraction of the 8 TB/s
HBM peak -- and the one with the most device time is named `dominant_kernel`; with the timed ceiling of
tools/atomic_peak.hip (device-scope 4-byte atomics per second, addresses over every channel) its atomics rate is set
against that ceiling: `roofline_atomics`.

A "launch" of the merge pass is one unit of the training loop = the a != b kernel(s) of a general iteration, a lean
iteration or a chain step plus, in the general path, the a == b kernel; bench.py times exactly that group.  Units =
launches of the table-update kernels (k_apply2, k_apply_lean, k_apply_chain; one per unit).  "merges" = trains x merges
per train (trains = launches of the first-statistics kernel).
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RD, WR, AT = "TCC_EA0_RDREQ_DRAM_32B", "TCC_EA0_WRREQ_WRITE_DRAM_32B", "TCC_EA0_WRREQ_WRITE_ATOMIC_32B"


def table(path):
    cur = sqlite3.connect(path).cursor()
    out = {}
    seen = {}
    for name, cn, v, d in cur.execute("select name, counter_name, counter_value, dispatch_id from pmc_events"):
        key = name.split("(")[0].replace("void ", "").replace("bpe::", "")
        # (the kernels that know the slot geometry exist twice: bpe_g4 = 1024-id slots, bpe_g1 = 256-id slots, marked "@256")
        key = key.replace("bpe_g4::", "") if "bpe_g1::" not in key else key.replace("bpe_g1::", "") + "@256"
        e = out.setdefault(key, {"calls": 0, RD: 0.0, WR: 0.0, AT: 0.0})
        if cn in e:
            e[cn] += float(v) * 32.0
        s = seen.setdefault(key, set())
        if d not in s:
            s.add(d)
            e["calls"] += 1
    return out


def short_name(name):
    key = name.split("(")[0].replace("void ", "").replace("bpe::", "")
    return key.replace("bpe_g4::", "") if "bpe_g1::" not in key else key.replace("bpe_g1::", "") + "@256"


def durations(csv_path):
    """kernel -> (calls, total_us) from tools/rocpd_stats.py's table"""
    import csv
    out = {}
    with open(csv_path, newline="") as fh:
        for row in csv.DictReader(fh):
            k = short_name(row["Name"])
            c, t = out.get(k, (0, 0.0))
            out[k] = (c + int(row["Calls"]), t + float(row["TotalDurationNs"]) / 1e3)
    return out


HBM_PEAK = 8.0e12


def main():
    db, workload, outp = sys.argv[1:4]
    n_in = int(sys.argv[4]) if len(sys.argv) > 4 else None
    merges_per_train = int(sys.argv[5]) if len(sys.argv) > 5 else None
    stats_csv = sys.argv[6] if len(sys.argv) > 6 and os.path.exists(sys.argv[6]) else None
    peak_json = sys.argv[7] if len(sys.argv) > 7 and os.path.exists(sys.argv[7]) else None
    t = table(db)
    kernels = {k: {"calls": v["calls"], "read_bytes": v[RD], "write_bytes": v[WR], "atomic_bytes": v[AT],
                   "hbm_bytes": v[RD] + v[WR] + v[AT]} for k, v in sorted(t.items())}
    merge = {k: v for k, v in kernels.items() if k.startswith("k_merge_")}
    units = sum(v["calls"] for k, v in kernels.items() if k.startswith(("k_apply2", "k_apply_lean", "k_apply_chain", "k_apply_delta")))
    trains = max(kernels.get("k_load_count", {}).get("calls", 0), kernels.get("k_pair_count_bytes", {}).get("calls", 0))
    total = sum(v["hbm_bytes"] for v in merge.values())
    total_all = sum(v["hbm_bytes"] for v in kernels.values())
    import bench
    first = kernels.get("k_load_count") or kernels.get("k_widen") or {}
    out = {
        "workload": workload, "source_hash": bench.source_hash(),
        "counters": [RD, WR, AT], "bytes": "32 x count (calibration: profiles/r4_pmc_calibration.json)",
        "launches": units, "trains": trains,
        "merges": trains * merges_per_train if merges_per_train else None,
        "hbm_bytes_total": total, "hbm_bytes_per_launch": total / units if units else None,
        "all_kernels_hbm_bytes_total": total_all,
        "all_kernels_hbm_bytes_per_launch": total_all / units if units else None,
        "merge_kernels": merge,
        "check_on_the_first_pass": {**first, "expected": "k_load_count: reads n input bytes + 8 B per chunk, writes 4n "
                                    "(k_widen: reads n, writes 4n)", "n_input_bytes": n_in},
        "other_kernels": {k: v for k, v in kernels.items() if not k.startswith("k_merge_")
                          and v["hbm_bytes"] > 0.002 * max(total, 1)},
    }
    if stats_csv:
        dur = durations(stats_csv)
        tab = {}
        for k, v in kernels.items():
            if k not in dur or not v["calls"]:
                continue
            calls_kt, total_us = dur[k]
            avg_us = total_us / max(calls_kt, 1)
            per = v["hbm_bytes"] / v["calls"]
            tab[k] = {"calls_pmc": v["calls"], "calls_trace": calls_kt, "total_us": round(total_us, 1), "avg_us": round(avg_us, 3),
                      "hbm_bytes_per_launch": round(per), "atomics_per_launch": round(v["atomic_bytes"] / 32.0 / v["calls"]),
                      "TBps": round(per / (avg_us * 1e-6) / 1e12, 4) if avg_us else None,
                      "frac_of_hbm_peak": round(per / (avg_us * 1e-6) / HBM_PEAK, 4) if avg_us else None,
                      "atomics_G_per_s": round(v["atomic_bytes"] / 32.0 / v["calls"] / (avg_us * 1e-6) / 1e9, 3) if avg_us else None}
        out["kernel_table"] = dict(sorted(tab.items(), key=lambda kv: -kv[1]["total_us"]))
        out["kernel_table_source"] = os.path.basename(stats_csv) + " (rocprofv3 --kernel-trace of the same command, its own run)"
        if tab:
            dk = max(tab, key=lambda k: tab[k]["total_us"])
            all_us = sum(t for _, t in dur.values())
            out["dominant_kernel"] = {"name": dk, "share_of_kernel_time": round(tab[dk]["total_us"] / all_us, 4), **tab[dk],
                                      "limited_by": None}
            if peak_json:
                with open(peak_json) as fh:
                    pk = json.load(fh)
                peak = pk["add_agent_spread_1GiB"]["G_per_s"]
                ach = tab[dk]["atomics_G_per_s"] or 0.0
                out["roofline_atomics"] = {
                    "kernel": dk, "achieved": ach, "peak": peak, "unit": "G atomics/s", "frac": round(ach / peak, 4) if peak else None,
                    "atomics_per_launch": tab[dk]["atomics_per_launch"], "avg_us": tab[dk]["avg_us"],
                    "peak_source": os.path.basename(peak_json) + " (tools/atomic_peak.hip, hipEvent-timed: non-returning device-scope "
                                   "atomicAdd, 4-byte words spread over 1 GiB, 256 x 1024 threads)",
                    "note": "atomics = TCC_EA0_WRREQ_WRITE_ATOMIC_32B of the kernel (a device-scope atomic is one 32-byte request at "
                            "the memory side, profiles/r4_pmc_calibration.json) / its launches / its average duration in the "
                            "kernel trace"}
                hbm = tab[dk]["frac_of_hbm_peak"] or 0.0
                out["dominant_kernel"]["limited_by"] = (
                    "hbm bytes" if hbm >= max(0.4, ach / peak if peak else 0) else
                    "instruction issue and per-slot latency, in balance (SQ counters of the same kernel, profiles/r6_sq_after_pass*.json: VALU pipes ~73 % busy at four waves per SIMD, waves parked ~58 % of their cycles), plus eight memory-side atomics per merge site -- not HBM bytes")
    with open(outp, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: out[k] for k in ("workload", "source_hash", "launches", "trains", "merges", "hbm_bytes_per_launch")}))
    print("first pass:", out["check_on_the_first_pass"])
    for k, v in merge.items():
        print(k, v)


if __name__ == "__main__":
    main()
