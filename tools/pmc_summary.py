#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) of the same bench command to HBM bytes per launch
of the dominant kernel.  gfx950 correction from the guide, re-checked on k_widen
(reads 100 MB, writes 400 MB): FETCH_SIZE reports half the bytes of wide coalesced
reads -> x2; WRITE_SIZE is accurate.
usage: tools/pmc_summary.py fetch.db write.db kernel_substring [out.json]"""
import json, sqlite3, sys
import numpy as np

def vals(path, counter, key):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, counter_value from pmc_events where counter_name=? order by dispatch_id",
                       (counter,)).fetchall()
    return np.array([v for n, v in rows if key in n]) * 1024.0

fdb, wdb, key = sys.argv[1:4]
f = vals(fdb, "FETCH_SIZE", key) * 2.0
w = vals(wdb, "WRITE_SIZE", key)
cal_f = vals(fdb, "FETCH_SIZE", "k_widen") * 2.0
cal_w = vals(wdb, "WRITE_SIZE", "k_widen")
out = {
    "kernel": key, "launches": int(len(f)),
    "fetch_bytes_per_launch": float(f.mean()), "write_bytes_per_launch": float(w.mean()),
    "hbm_bytes_per_launch": float(f.mean() + w.mean()),
    "calibration_k_widen": {"fetch_x2_bytes": float(cal_f[0]) if len(cal_f) else None,
                            "write_bytes": float(cal_w[0]) if len(cal_w) else None,
                            "expected": "reads n bytes, writes 4n"},
    "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE x1",
}
print(json.dumps(out, indent=1))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
