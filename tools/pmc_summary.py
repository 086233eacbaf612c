#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) of the same bench command to the HBM bytes the merge pass moves.

gfx950 correction from the guide, re-checked here on k_widen (reads n bytes with 16 B per lane,
writes 4n): FETCH_SIZE reports half the bytes of wide coalesced reads -> x2; WRITE_SIZE x1.

usage: tools/pmc_summary.py fetch.db write.db workload_name out.json [n_bytes_of_the_input]

A "launch" of the merge pass is one training iteration = the a != b kernel (dense, sparse or lean)
plus, in the general path, the a == b kernel (a no-op unless the pair has a == b); bench.py times
exactly that group.  The number of iterations = launches of the table-update kernel (k_apply2 in the
general path, k_apply_lean in lean iterations; one per iteration).  "all_kernels" = every kernel of
the run, for the whole-iteration fraction.
"""
import json
import os
import sqlite3
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def table(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, counter_value from pmc_events where counter_name=?", (counter,)).fetchall()
    out = {}
    for name, v in rows:
        key = name.split("(")[0].replace("void ", "").replace("bpe::", "")
        d = out.setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += float(v) * 1024.0  # the counters are in KiB
    return out


def main():
    fdb, wdb, workload, outp = sys.argv[1:5]
    n_in = int(sys.argv[5]) if len(sys.argv) > 5 else None
    f = table(fdb, "FETCH_SIZE")
    w = table(wdb, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        calls = max(f.get(k, [0])[0], w.get(k, [0])[0])
        kernels[k] = {"calls": calls, "fetch_bytes_x2": 2.0 * f.get(k, [0, 0.0])[1], "write_bytes": w.get(k, [0, 0.0])[1]}
    merge = {k: v for k, v in kernels.items() if k.startswith("k_merge_")}
    iters = sum(v["calls"] for k, v in kernels.items() if k.startswith(("k_apply2", "k_apply_lean", "k_apply_delta")))
    if not iters:
        iters = max((v["calls"] for k, v in merge.items() if k.startswith("k_merge_aa")), default=0)
    total = sum(v["fetch_bytes_x2"] + v["write_bytes"] for v in merge.values())
    total_all = sum(v["fetch_bytes_x2"] + v["write_bytes"] for v in kernels.values())
    import bench
    out = {
        "workload": workload, "source_hash": bench.source_hash(),
        "launches": iters, "hbm_bytes_total": total,
        "hbm_bytes_per_launch": total / iters if iters else None,
        "all_kernels_hbm_bytes_total": total_all,
        "all_kernels_hbm_bytes_per_iteration": total_all / iters if iters else None,
        "merge_kernels": merge,
        "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE x1; counters are KiB",
        "calibration_k_widen": {**kernels.get("k_widen", {}), "expected": "reads n input bytes, writes 4n",
                                "n_input_bytes": n_in},
        "other_kernels": {k: v for k, v in kernels.items() if not k.startswith("k_merge_") and k != "k_widen"
                          and v["fetch_bytes_x2"] + v["write_bytes"] > 0.002 * max(total, 1)},
    }
    with open(outp, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: out[k] for k in ("workload", "source_hash", "launches", "hbm_bytes_per_launch")}))
    print("calibration k_widen:", out["calibration_k_widen"])
    for k, v in merge.items():
        print(k, v)


if __name__ == "__main__":
    main()
