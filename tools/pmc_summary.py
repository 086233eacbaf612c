#!/usr/bin/env python3
"""Reduce ONE rocprofv3 --pmc pass of a bench command to the HBM bytes its kernels move.

Counters (gfx950, raw, size-weighted: a 64-byte request counts 2, a 128-byte one 4):
    TCC_EA0_RDREQ_DRAM_32B  TCC_EA0_WRREQ_WRITE_DRAM_32B  TCC_EA0_WRREQ_WRITE_ATOMIC_32B
bytes = 32 x count.  Calibrated on kernels with known byte counts in this engine's access patterns
(tools/pmc_calib.hip, profiles/r4_pmc_calibration.json): reads 1.00 x requested for wide and narrow coalesced streams,
1.05 x for the merge pass's slot + header + mask pattern; every read request is 128 bytes (a random 32-byte record
costs 128, a 4-byte gather 128); writes exact; a device atomic counts as one 32-byte write.  The derived FETCH_SIZE
tallies a 128-byte request at 64 (x2 for EVERY read pattern, not only wide streams), WRITE_SIZE = writes + atomics.

usage: tools/pmc_summary.py raw.db workload_name out.json [n_bytes_of_the_input]

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


def main():
    db, workload, outp = sys.argv[1:4]
    n_in = int(sys.argv[4]) if len(sys.argv) > 4 else None
    merges_per_train = int(sys.argv[5]) if len(sys.argv) > 5 else None
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
    with open(outp, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: out[k] for k in ("workload", "source_hash", "launches", "trains", "merges", "hbm_bytes_per_launch")}))
    print("first pass:", out["check_on_the_first_pass"])
    for k, v in merge.items():
        print(k, v)


if __name__ == "__main__":
    main()
