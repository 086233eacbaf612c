#!/usr/bin/env python3
"""HBM bytes of the encode batches of `bench.py --workload encode --steps 1 --warmup 0` from ONE rocprofv3 --pmc pass
(the raw counters of tools/pmc_summary.py).  A batch = the dispatches from one k_enc_tab_init to the next; the bench
encodes with the cfg3 table first, then with the cl100k-sized one.
usage: tools/pmc_encode_summary.py raw.db out.json"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RD, WR, AT = "TCC_EA0_RDREQ_DRAM_32B", "TCC_EA0_WRREQ_WRITE_DRAM_32B", "TCC_EA0_WRREQ_WRITE_ATOMIC_32B"


def main():
    db, outp = sys.argv[1:3]
    cur = sqlite3.connect(db).cursor()
    disp = {}
    for name, cn, v, d, start in cur.execute("select name, counter_name, counter_value, dispatch_id, start from pmc_events"):
        e = disp.setdefault(d, {"name": name.split("(")[0].replace("void ", "").replace("bpe::", ""), "start": start, RD: 0.0, WR: 0.0, AT: 0.0})
        if cn in e:
            e[cn] += float(v) * 32.0
    seq = sorted(disp.values(), key=lambda e: e["start"])
    batches, cur_b = [], None
    for e in seq:
        if e["name"].startswith("k_enc_tab_init"):
            cur_b = {}
            batches.append(cur_b)
        if cur_b is None or e["name"].startswith(("k_merge_chain", "k_chain_sel", "k_apply_chain", "k_select", "k_load_count")):
            if e["name"].startswith(("k_load_count", "k_chain_sel")):
                cur_b = None  # (a training run follows: not part of an encode batch)
            continue
        k = cur_b.setdefault(e["name"], {"calls": 0, "read_bytes": 0.0, "write_bytes": 0.0, "atomic_bytes": 0.0})
        k["calls"] += 1
        k["read_bytes"] += e[RD]
        k["write_bytes"] += e[WR]
        k["atomic_bytes"] += e[AT]
    import bench
    names = ["cfg3", "cl100k_sized"]
    out = {"source_hash": bench.source_hash(), "counters": [RD, WR, AT], "bytes": "32 x count (profiles/r4_pmc_calibration.json)",
           "batches": len(batches), "hbm_bytes_per_step": {}, "kernels": {}}
    for nm, b in zip(names, batches[-2:] if len(batches) >= 2 else batches):
        out["kernels"][nm] = b
        out["hbm_bytes_per_step"][nm] = sum(k["read_bytes"] + k["write_bytes"] + k["atomic_bytes"] for k in b.values())
    with open(outp, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({"batches": len(batches), "hbm_bytes_per_step": out["hbm_bytes_per_step"]}))


if __name__ == "__main__":
    main()
