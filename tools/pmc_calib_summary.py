#!/usr/bin/env python3
"""Join tools/pmc_calib's requested byte counts with the counters of its three rocprofv3 runs.
usage: tools/pmc_calib_summary.py requested.json fetch_dir write_dir raw_dir  > calibration.json"""
import glob
import json
import os
import sqlite3
import sys


def counters(d):
    """{kernel: {counter: [sum of values, dispatches]}} of the first rocpd database under d ({} if there is none)."""
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        return {}
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    disp = next((c for c in ("dispatch_id", "event_id", "id") if c in cols), None)
    q = f"select name, counter_name, counter_value, {disp or '0'} from pmc_events"
    out = {}
    seen = {}
    for name, cn, v, d_id in cur.execute(q):
        k = name.split("(")[0].replace("void ", "")
        e = out.setdefault(k, {}).setdefault(cn, [0.0, 0])
        e[0] += float(v)
        s = seen.setdefault((k, cn), set())
        if d_id not in s:
            s.add(d_id)
            e[1] += 1
    out["_columns"] = cols
    return out


def main():
    req = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f, w, r = (counters(p) for p in sys.argv[2:5])
    n = req["launches_each"]
    res = {"launches_each": n, "pmc_events_columns": r.get("_columns") or f.get("_columns"), "kernels": {}}
    for k, want in req["requested_bytes_per_launch"].items():
        row = {"pattern": want["pattern"], "requested_read": want["read"], "requested_write": want["write"]}
        def per_launch(tab, cn, scale):
            e = tab.get(k, {}).get(cn)
            return None if not e else e[0] * scale / n
        row["FETCH_SIZE_bytes"] = per_launch(f, "FETCH_SIZE", 1024.0)
        row["WRITE_SIZE_bytes"] = per_launch(w, "WRITE_SIZE", 1024.0)
        row["RDREQ_DRAM_32B_x32"] = per_launch(r, "TCC_EA0_RDREQ_DRAM_32B", 32.0)
        row["WRREQ_WRITE_DRAM_32B_x32"] = per_launch(r, "TCC_EA0_WRREQ_WRITE_DRAM_32B", 32.0)
        row["WRREQ_WRITE_ATOMIC_32B_x32"] = per_launch(r, "TCC_EA0_WRREQ_WRITE_ATOMIC_32B", 32.0)
        row["RDREQ_128B_count"] = per_launch(r, "TCC_EA0_RDREQ_128B", 1.0)
        for key, base in (("FETCH_SIZE_bytes", "requested_read"), ("RDREQ_DRAM_32B_x32", "requested_read"),
                          ("WRITE_SIZE_bytes", "requested_write"), ("WRREQ_WRITE_DRAM_32B_x32", "requested_write")):
            if row[key] is not None and row[base]:
                row[key + "_over_requested"] = round(row[key] / row[base], 4)
        res["kernels"][k] = row
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
