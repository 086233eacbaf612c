#!/usr/bin/env python3
"""Where a one-launch chain step (k_step.hip) spends its time: reduce the clock stamps the library dumps when
BPE_STEP_STAMPS=<file> is set (100 MHz clock; [step % 8192][workgroup 0 | gm / 2 | last][8 stamps]) to medians per
phase of training.
    BPE_STEP_STAMPS=gpurun_out/stamps.bin python tools/train_n.py regex1g 31744 && python tools/step_stamps.py gpurun_out/stamps.bin
stamps: 0 entry | 1 selection done (workgroup 0) | 2 batch known (published / received) | 3 merge pass done |
4 grid barrier passed | 5 table update (tokens) done | 6 records done | 7 headers committed (end)
workgroup 0, inside the selection: 8 state read | 9 maintain done (table words, compaction) | 10 rebuild or not decided | 11 sorted |
12 levels analysed | 13 located | 14 finished;  other workgroups, inside the merge pass: 8 hash table built | 9 first candidate list built"""
import json
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.int64)
RING = 8192
a = raw[:RING * 3 * 16].reshape(RING, 3, 16)
per_wg = raw[RING * 3 * 16:].reshape(RING, 256, 2) if len(raw) > RING * 3 * 16 else None
live = np.flatnonzero(a[:, 0, 0] != 0)
out = {"steps_with_stamps": int(len(live)), "unit": "us (100 MHz clock)", "ranges": []}
names = ["sel", "publish/receive", "merge", "barrier", "tokens", "records", "commit"]
if len(live):
    lo, hi = int(live.min()), int(live.max()) + 1
    edges = np.linspace(lo, hi, 9).astype(int)
    for i in range(8):
        idx = [s for s in range(edges[i], edges[i + 1]) if a[s, 0, 0] != 0 and a[s, 0, 7] != 0]
        if not idx:
            continue
        r = {"steps": [int(edges[i]), int(edges[i + 1])], "n": len(idx)}
        for w, wn in enumerate(("wg0", "wg_mid", "wg_last")):
            d = {}
            t0 = a[idx, 0, 0]  # everything relative to workgroup 0's entry
            for k in range(16):
                v = a[idx, w, k]
                ok = v != 0
                if ok.any():
                    d[f"t{k}"] = round(float(np.median((v - t0)[ok])) / 100.0, 2)
            r[wn] = d
        r["total_wg0_us"] = round(float(np.median(a[idx, 0, 7] - a[idx, 0, 0])) / 100.0, 2)
        if per_wg is not None:
            t0 = a[idx, 0, 0][:, None]
            got, arr = per_wg[idx, :, 0], per_wg[idx, :, 1]
            ok = (got != 0) & (arr != 0)
            rel_arr = np.where(ok, arr - t0, -1)
            last = rel_arr.argmax(axis=1)
            r["barrier_arrival"] = {
                "last_arrival_us_median": round(float(np.median(rel_arr.max(axis=1))) / 100.0, 2),
                "median_arrival_us": round(float(np.median(rel_arr[ok])) / 100.0, 2),
                "last_is_wg_1_to_63_share": round(float(((last >= 1) & (last <= 63)).mean()), 3),
                "merge_us_median_wg_1_63": round(float(np.median((arr - got)[:, 1:64][ok[:, 1:64]])) / 100.0, 2),
                "merge_us_median_wg_64_255": round(float(np.median((arr - got)[:, 64:][ok[:, 64:]])) / 100.0, 2),
                "merge_us_p90_all": round(float(np.percentile((arr - got)[ok], 90)) / 100.0, 2),
                "merge_us_max_median": round(float(np.median(np.where(ok, arr - got, 0).max(axis=1))) / 100.0, 2),
                "line_received_us_median": round(float(np.median((got - t0)[ok])) / 100.0, 2),
                "line_received_us_last_median": round(float(np.median(np.where(ok, got - t0, 0).max(axis=1))) / 100.0, 2)}
        out["ranges"].append(r)
print(json.dumps(out, indent=1))
