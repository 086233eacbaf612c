#!/usr/bin/env python3
"""bench.py -- headline benchmark (BASELINE.json: "BPE merges/sec + pair-count
GB/s vs HBM roofline").

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME] [--secondary a,b]

Workloads (one "step" = one complete train() from bytes already resident in HBM:
widen to ids, initial get_stats, then vocab-256 x (arg-max with the reference's
tie-break, merge, pair-table update)):

  regex1g  (default, N=1)  BASELINE.json configs[2]: RegexTokenizer.train with the
           GPT-4 split pattern (regex.py:19,41) on 1 GB of synthetic UTF-8
           (synth_text(1e9, seed 2)), vocab 32000 = 31,744 merges, no chunk
           de-duplication.  The split runs on the host (bpe_split) before the clock.
  basic1g  the north_star target sentence: BasicTokenizer.train, same 1 GB, vocab 32000.
  cfg2     BASELINE.json configs[1]: BasicTokenizer.train, 100 MB (seed 1), vocab 4096.
  encode   BASELINE.json configs[4] shape: batch encode of documents (own vocabulary).

The headline workload is timed for --steps; the --secondary workloads (default
basic1g,cfg2,regex1g_dedup,e2e_class,literal_path,encode,encode_long at N=1) run --secondary-steps each after it and are reported under
"secondary".  N > 1: the chunk list of regex1g is sharded (contiguous chunk ranges,
--bytes per GPU = cfg4 shape, weak scaling); `value` = what ALL ranks processed per second
(N x the job's merges/s: a merge of the one sharded job is applied to every rank's shard --
the bench contract's whole-job aggregate), `job_merges_per_s` = the rate of the ONE job.

Prints ONE JSON line on rank 0.  `roofline` = the dominant kernel class timed with
hipEvents on the library's own stream inside the timed steps; `cpu_baseline` = the CPU
oracle (C restatement of the reference loop, one thread) on a bounded slice of the same
input, on this host, plus minbpe's pure-Python loop (oracle/pyref.py, a pinned restatement) on a smaller slice.
Parity with the oracle is CHECKED in the run against committed full-length digests
(tests/golden/big_golden.json) and reported, never assumed.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s)
HBM_COPY_GBPS = 6290.0  # measured float4 copy, same guide
PROFILE_ROUND = "r6"      # prefix of the committed rocprofv3 summaries under profiles/ this line attaches
PY_SAMPLE_BYTES = 16_000_000  # cpu_baseline: the pure-Python loop is timed on this much of the input ...
PY_SAMPLE_MERGES = 4          # ... for this many merges (~10 s of one host core)

WORKLOADS = {
    "regex1g": dict(bytes=1_000_000_000, seed=2, vocab=32000, chunked=True,
                    desc="RegexTokenizer.train (GPT-4 split pattern, no de-duplication)"),
    "basic1g": dict(bytes=1_000_000_000, seed=2, vocab=32000, chunked=False, desc="BasicTokenizer.train"),
    "cfg2": dict(bytes=100_000_000, seed=1, vocab=4096, chunked=False, desc="BasicTokenizer.train"),
    # the headline input trained on its DISTINCT chunks with multiplicities (RegexTokenizer.dedup):
    # same merges and counts, reported next to the headline, never instead of it
    "regex1g_dedup": dict(bytes=1_000_000_000, seed=2, vocab=32000, chunked=True, dedup=True,
                          desc="RegexTokenizer.train (GPT-4 split pattern) on de-duplicated chunks"),
}


HOST_ONLY_SOURCES = ("synth.cpp", "split.cpp", "dedup.cpp", "unicode_tables.h", "utf8.cpp")


def source_hash():
    """sha256 over the sources that make the kernels and their launches (every .hip / .h / .cpp under minbpe_amd/csrc
    except the host-only translation units: the text generator, the pre-split scanner with its tables, the chunk
    de-duplication -- no device code, nothing a kernel's counters depend on): identifies the kernels a committed PMC
    profile was measured on."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "minbpe_amd", "csrc")
    for d, _, files in sorted(os.walk(base)):
        for f in sorted(files):
            if f.endswith((".hip", ".h", ".cpp")) and f not in HOST_ONLY_SOURCES:
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def host_info():
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def golden_entry(name):
    p = os.path.join(ROOT, "tests", "golden", "big_golden.json")
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f).get(name)


def all_golden_names():
    p = os.path.join(ROOT, "tests", "golden", "big_golden.json")
    if not os.path.exists(p):
        return []
    with open(p) as f:
        return sorted(json.load(f).keys())


def parity_report(name, wl, data_sha, offs, res):
    """Compare the GPU result with the committed oracle digests of this exact input: the plain oracle's
    (the reference's own loop, as far as hours of CPU reach) and, where it exists, the weighted oracle's
    (`<name>_w`: orc_train_weighted on the distinct chunks -- the WHOLE merge list; its first merges are
    checked against the plain oracle's when the fixture is generated)."""
    from helpers import checkpoint_digests, first_divergence
    rep = {"golden": None, "merges_checked": 0, "equal": None}
    # the entries named after the workload first, then any other entry of this exact input (size, seed, chunked or
    # not, sha256 of the bytes): e.g. `--bytes 3900000000` finds regex3p9g_w
    names = [name, name + "_w"] + [k for k in all_golden_names() if k not in (name, name + "_w")]
    for gname in names:
        g = golden_entry(gname)
        if (not g or g.get("world") or g["bytes"] != wl["bytes"] or g["seed"] != wl["seed"]
                or g.get("data_sha256") != data_sha or bool(g.get("chunked")) != bool(wl.get("chunked"))):
            continue  # (entries with a "world" are sharded jobs: the N > 1 leg's, per-shard sha256)
        if g.get("offsets_sha256") and offs is not None:
            same_split = hashlib.sha256(offs.tobytes()).hexdigest() == g["offsets_sha256"]
            rep["split_equals_regex_module"] = same_split
            if not same_split:
                continue
        k = min(g["done"], len(res["pairs"]))
        got = checkpoint_digests(res["pairs"][:k], res["counts"][:k], res["lens"][:k], g["step"])
        bad = first_divergence(got, g["digests"])
        checked = [c for c, _ in got if c in {c2 for c2, _ in g["digests"]}]
        n_ok = max(checked) if checked else 0
        kind = ("weighted oracle on the distinct chunks" if g.get("weighted") else
                "incremental exact trainer, pinned to the plain oracle" if g.get("fast") else "oracle")
        rep.setdefault("goldens", []).append(
            {"entry": f"tests/golden/big_golden.json[{gname}] ({kind}, {g['done']} merges)",
             "merges_checked": n_ok, "equal": bool(checked) and bad is None})
        if bad is not None:
            rep["first_bad_checkpoint"] = min(bad, rep.get("first_bad_checkpoint", bad))
        rep["equal"] = (rep["equal"] is not False) and bool(checked) and bad is None
        if n_ok >= rep["merges_checked"]:
            rep["merges_checked"] = n_ok
            rep["golden"] = rep["goldens"][-1]["entry"]
    return rep


def invariants(res, n0):
    """Size-independent checks at full length: every a != b merge removes exactly `count` ids;
    an a == b merge removes between ceil(count/2)... and count; counts never increase."""
    import numpy as np
    lens = np.array([n0] + res["lens"], dtype=np.int64)
    cnt = np.array(res["counts"], dtype=np.int64)
    pairs = np.array(res["pairs"], dtype=np.int64).reshape(-1, 2)
    same = pairs[:, 0] == pairs[:, 1]
    removed = lens[:-1] - lens[1:]
    ok = bool(np.all(removed[~same] == cnt[~same]) and np.all(removed[same] <= cnt[same])
              and np.all(removed > 0) and np.all(np.diff(cnt) <= 0))
    return {"len_drop_equals_count_and_counts_monotone": ok, "merges": int(len(cnt))}


_INPUT_CACHE = {}


def synth_cached(n, seed):
    """synth_text is a sequential generator (~30 MB/s): keep the last stream in memory and, for
    streams of 100 MB and more, on local disk, so that several workloads / tools of one session
    do not regenerate it."""
    import minbpe_amd
    if (n, seed) in _INPUT_CACHE:
        return _INPUT_CACHE[(n, seed)]
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"minbpe_synth_{n}_{seed}.bin")
    data = None
    if n >= 100_000_000 and os.path.exists(path) and os.path.getsize(path) == n:
        with open(path, "rb") as f:
            data = f.read()
    if data is None:
        data = minbpe_amd.synth_text(n, seed)
        if n >= 100_000_000:
            try:
                with open(path + ".tmp", "wb") as f:
                    f.write(data)
                os.replace(path + ".tmp", path)
            except OSError:
                pass
    _INPUT_CACHE.clear()
    _INPUT_CACHE[(n, seed)] = data
    return data


def make_input(wl, rank=0):
    from minbpe_amd import _native
    t0 = time.perf_counter()
    data = synth_cached(wl["bytes"], wl["seed"] + rank)
    offs = _native.split_offsets(data, 4) if wl["chunked"] else None
    return data, offs, time.perf_counter() - t0


def run_train_workload(name, wl, eng, steps, warmup, barrier, reduce_max, mode, secondary=False):
    """Upload once, warm up, one untimed step with events around every kernel class (the
    breakdown), then `steps` timed steps with events around the merge pass only."""
    data, offs, prep_s = make_input(wl)
    data_sha = hashlib.sha256(data).hexdigest()
    num_merges = wl["vocab"] - 256
    if mode >= 0:
        eng.set_option("mode", mode)
    t0 = time.perf_counter()
    eng.load_bytes(data, offs)  # H2D once, outside the timed region
    upload_s = time.perf_counter() - t0
    step = lambda: eng.train(num_merges)
    for _ in range(warmup):
        step()
    eng.set_option("profile", 2)  # (hipEvents around every kernel class; sampled like profile 1, see below)
    eng.prof_reset()
    step()
    breakdown = eng.prof_read()
    eng.set_option("profile", 1)
    eng.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    barrier()
    dt = reduce_max(time.perf_counter() - t0)
    prof = eng.prof_read()
    eng.set_option("profile", 0)
    # the same step without the event records of the timed region (what they cost: an event drains the queue)
    t0 = time.perf_counter()
    step()
    plain_ms = (time.perf_counter() - t0) * 1e3
    out = {
        "workload": f"{wl['desc']}, {wl['bytes']} B synthetic UTF-8 (seed {wl['seed']}), vocab {wl['vocab']} "
                    f"({num_merges} merges)" + (f", {len(offs)} chunks" if offs is not None else ""),
        "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
        "merges_per_s": round(num_merges * steps / dt, 2),
        "ms_per_step_without_event_records": round(plain_ms, 3),
        "host_prep_s": round(prep_s, 2), "upload_s": round(upload_s, 3),
        "pcie_inclusive_merges_per_s": round(num_merges / (dt / steps + upload_s), 2),
        "final_len": res["lens"][-1] if res["lens"] else len(data),
        "parity": parity_report(name, wl, data_sha, offs, res),
        "invariants": invariants(res, len(data)),
    }
    # dominant kernel class by device time -> roofline (timed live in the timed region: hipEvents on the
    # library's stream around that class, every iteration up to 1024 and every 64th after, weighted)
    hot = max(("pair_count", "merge", "widen"), key=lambda k: breakdown[k]["ms"])
    hp = prof[hot] if prof[hot]["launches"] else breakdown[hot]
    alg_GBps = hp["alg_bytes"] / (hp["ms"] * 1e-3) / 1e9 if hp["ms"] > 0 else 0.0
    launches = max(hp["launches"], 1)
    avg_launch_s = hp["ms"] / launches * 1e-3
    roofline = {
        "bound": "hbm",
        "kernel": {"merge": "merge pass = k_merge_chain | k_merge_chain_dense | k_merge_chain_dense1 (a chain step: 1..15 "
                            "merges + their pair-table deltas in one sweep) | k_merge_ab_* + k_merge_aa (a general iteration)",
                   "pair_count": ("get_stats of every iteration: k_load_count (the first, on bytes) then the general histogram "
                                  "k_pair_count_h32 / k_pair_count_lds (recount mode)" if mode == 0 else
                                  "k_load_count (bytes -> ids + chunk starts + pair counts)"), "widen": "k_widen"}[hot],
        "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
        "traffic": None, "traffic_source": None,
        "launches": hp["launches"], "avg_launch_ms": round(avg_launch_s * 1e3, 5),
        "alg_bytes_per_launch": hp["alg_bytes"] // launches,
        "equivalent_work_GBps": round(alg_GBps, 1), "equivalent_work_frac": round(alg_GBps / HBM_PEAK_GBPS, 4),
        "note": "achieved / frac = PHYSICAL, per train on both sides: HBM bytes of the merge-pass kernels per train (32 x the "
                "size-weighted TCC/EA request counters of a rocprofv3 --pmc pass of this same command -- calibrated on known "
                "byte counts, profiles/r4_pmc_calibration.json -- committed profile, attached only when it was measured on "
                "these library sources) / the hipEvent time of the merge passes per train / 8 TB/s; `traffic` = the same "
                "bytes per launch this run counted.  The pass is NOT bound by HBM bytes once it is sparse: see "
                "`limited_by`, `dominant_kernel` and the line's `roofline_atomics`.  equivalent_work_* = the SURVEY 8d ALGORITHMIC bytes, 4(2N_i + "
                "N_{i+1}) per merge (what the reference's get_stats + merge touch), over the same time: the pass does "
                "not re-read the stream for get_stats and skips slots a merge cannot touch, so that figure exceeds the "
                "physical one (and 1) -- it measures work avoided, not the kernel.",
    }
    # HBM bytes per launch: PMC counters cannot be read from inside this process; they come from the
    # committed rocprofv3 --pmc passes of this same command, and only if they were measured on the
    # same library sources (hash below).
    whole = {"device_ms_per_train": None, "device_ms_per_iteration": None, "traffic": None, "frac": None}
    dev_ms = sum(v["ms"] for v in breakdown.values())
    whole["device_ms_per_train"] = round(dev_ms, 3)
    whole["device_ms_per_iteration"] = round(dev_ms / num_merges, 5)
    roofline_atomics = None
    pmc_file = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{name}_pmc.json")
    if os.path.exists(pmc_file):
        with open(pmc_file) as f:
            pmc = json.load(f)
        if pmc.get("source_hash") == source_hash() and pmc.get("launches") and pmc.get("trains"):
            src = f"profiles/{PROFILE_ROUND}_{name}_pmc.json (source_hash {pmc['source_hash']})"
            if hot == "merge":
                # PER TRAIN on both sides: the profile's bytes of the merge-pass kernels per train over the hipEvent time
                # of the merge passes per train; the per-launch figures divide both by the launches THIS run counted
                trains_timed = steps if prof[hot]["launches"] else 1
                per_train = pmc["hbm_bytes_total"] / pmc["trains"]
                pass_s_per_train = hp["ms"] * 1e-3 / trains_timed
                launches_per_train = max(hp["launches"] / trains_timed, 1)
                roofline["traffic"] = int(per_train / launches_per_train)
                roofline["traffic_per_train"] = int(per_train)
                roofline["pass_ms_per_train"] = round(pass_s_per_train * 1e3, 3)
                roofline["traffic_source"] = src
                roofline["achieved"] = round(per_train / pass_s_per_train / 1e9, 1)
                roofline["frac"] = round(per_train / pass_s_per_train / 1e9 / HBM_PEAK_GBPS, 4)
                # the dominant KERNEL by itself (the class above is eight kernels): its own bytes and atomics per launch
                # from the PMC pass, its own average duration from the kernel trace of the same command
                dk = pmc.get("dominant_kernel")
                if dk:
                    roofline["dominant_kernel"] = dk
                    roofline["limited_by"] = dk.get("limited_by")
                roofline_atomics = pmc.get("roofline_atomics")
            if pmc.get("all_kernels_hbm_bytes_total"):
                # per TRAIN on both sides (the profile's per-launch figure is per unit of the loop -- a chain step does
                # several merges -- and must not be set against a per-merge time)
                per_train = pmc["all_kernels_hbm_bytes_total"] / pmc["trains"]
                whole["traffic_per_train"] = int(per_train)
                whole["traffic"] = int(per_train / num_merges)
                whole["frac"] = round(per_train / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                whole["traffic_source"] = src
        else:
            roofline["traffic_source"] = "committed PMC profile is from other library sources: not attached"
    out["roofline_atomics"] = roofline_atomics
    out["whole_iteration"] = whole
    roofline["achieved_kind"] = "physical (PMC traffic / hipEvent time)"
    if roofline["achieved"] is None and secondary:
        # a secondary workload without a PMC pass of these sources: no roofline fraction is claimed for it (the
        # algorithmic figure stays under equivalent_work_*: work avoided, not a fraction of anything)
        roofline["achieved_kind"] = "none: no PMC profile of this workload on these sources (see equivalent_work_*)"
    elif roofline["achieved"] is None:  # no PMC profile of these sources: the algorithmic figure, labelled as such
        roofline["achieved"] = round(alg_GBps, 1)
        roofline["frac"] = round(alg_GBps / HBM_PEAK_GBPS, 4)
        roofline["achieved_kind"] = "algorithmic (SURVEY 8d bytes / hipEvent time): no PMC profile of these sources"
    out["roofline"] = roofline
    pc, mg = breakdown["pair_count"], breakdown["merge"]
    alg_bytes_step = pc["alg_bytes"] + mg["alg_bytes"]
    out["pair_count_GBps"] = round(pc["alg_bytes"] / (pc["ms"] * 1e-3) / 1e9, 1) if pc["ms"] else None
    out["merge_GBps"] = round(mg["alg_bytes"] / (mg["ms"] * 1e-3) / 1e9, 1) if mg["ms"] else None
    out["iter_GBps_wall"] = round(alg_bytes_step * steps / dt / 1e9, 1)
    out["device_ms_per_step"] = {k: round(v["ms"], 3) for k, v in breakdown.items() if v["ms"]}
    out["merge_passes"] = eng.train_stats()  # dense / sparse passes, index builds of the last train()
    return out, data, offs, res


def run_dedup_workload(wl, eng, steps, barrier, ref):
    """Chunk de-duplication on the host (bpe_dedup_chunks), weighted upload, `steps` timed trains.
    ref = (pairs, counts) of the plain run on the same input, when the headline produced them."""
    from minbpe_amd import _native
    data, offs, prep_s = make_input(wl)
    num_merges = wl["vocab"] - 256
    t0 = time.perf_counter()
    d2, o2, w, nd = _native.dedup_chunks(data, offs)
    dedup_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.load_bytes(d2, o2, w)
    upload_s = time.perf_counter() - t0
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = eng.train(num_merges)
    barrier()
    dt = (time.perf_counter() - t0) / steps
    same = None
    if ref is not None:
        same = bool(res["pairs"] == ref[0] and res["counts"] == ref[1])
    return {
        "workload": f"{wl['desc']}, {wl['bytes']} B synthetic UTF-8 (seed {wl['seed']}), vocab {wl['vocab']} "
                    f"({num_merges} merges): {len(offs)} chunks -> {nd} distinct -> {len(o2)} weighted chunks, "
                    f"{len(d2)} B on the device",
        "steps": steps, "ms_per_step": round(dt * 1e3, 3), "merges_per_s": round(num_merges / dt, 2),
        "dedup_host_s": round(dedup_s, 3), "upload_s": round(upload_s, 3), "host_prep_s": round(prep_s, 2),
        "end_to_end_merges_per_s": round(num_merges / (dt + dedup_s + upload_s), 2),
        "same_merges_and_counts_as_plain_run": same,
        "counts_monotone": bool(all(res["counts"][i] >= res["counts"][i + 1] for i in range(len(res["counts"]) - 1))),
        "merge_passes": eng.train_stats(),
    }


def run_literal_path_workload(wl, eng, ref, nm=192):
    """The path exactly as the reference walks it (mode=0, basic.py:31-42 / regex.py:49-63): EVERY iteration a full
    get_stats over the stream (k_load_count for the first, then the general LDS-cached histogram), arg-max, a full merge
    (three passes: tile summaries, their scan, the rewrite into the other buffer) -- no incremental counts, no slots, no
    index.  The first `nm` merges of the headline input, timed with hipEvents around every kernel class (profile 2),
    against SURVEY 8d's algorithmic bytes of those iterations, B_i = 4 (2 N_i + N_{i+1}): get_stats 4 N_i, merge
    4 (N_i + N_{i+1}).  ref = (pairs, counts) of the headline run (themselves checked against the golden digests)."""
    data, offs, _ = make_input(wl)
    eng.set_option("mode", 0)
    try:
        eng.load_bytes(data, offs)
        eng.train(8)  # warm
        eng.set_option("profile", 2)
        eng.prof_reset()
        t0 = time.perf_counter()
        res = eng.train(nm)
        wall = time.perf_counter() - t0
        bd = eng.prof_read()
    finally:
        eng.set_option("profile", 0)
        eng.set_option("mode", 1)
    lens = [len(data)] + [int(x) for x in res["lens"]]
    b_stats = 4 * sum(lens[:-1])
    b_merge = 4 * sum(lens[i] + lens[i + 1] for i in range(nm))
    t = {k: bd[k]["ms"] * 1e-3 for k in ("pair_count", "argmax", "merge", "table")}
    t_iter = sum(t.values())
    frac = lambda b, sec: round(b / sec / 1e9 / HBM_PEAK_GBPS, 4) if sec > 0 else None
    same = None
    if ref is not None:
        same = bool(res["pairs"] == ref[0][:nm] and res["counts"] == ref[1][:nm])
    return {
        "workload": f"the reference's loop as written (mode=0: get_stats + max + merge over the whole stream every iteration), "
                    f"first {nm} merges of {wl['desc']}, {wl['bytes']} B synthetic UTF-8 (seed {wl['seed']})",
        "merges": nm, "same_merges_and_counts_as_headline_run": same,
        "merges_per_s": round(nm / wall, 1), "ms_per_iteration_wall": round(wall / nm * 1e3, 4),
        "ids_first": lens[0], "ids_last": lens[-1],
        "device_ms_per_iteration": {k: round(v / nm * 1e3, 4) for k, v in t.items()},
        "get_stats": {"alg_bytes": b_stats, "GBps": round(b_stats / t["pair_count"] / 1e9, 1), "frac_of_hbm_peak": frac(b_stats, t["pair_count"]),
                      "limited_by": "memory-side atomics, not HBM bytes: the flush of every workgroup's LDS table and the pairs the table "
                                    "had no slot for (profiles/r6_ap_pair_count_experiment.jsonl: the same grid reads the stream at 0.71 of "
                                    "the peak when it does nothing else, 0.40-0.43 with the LDS work, 0.08-0.26 with flush and misses)"},
        "merge": {"alg_bytes": b_merge, "GBps": round(b_merge / t["merge"] / 1e9, 1), "frac_of_hbm_peak": frac(b_merge, t["merge"]),
                  "note": "three passes move 4 (2 N_i + N_{i+1}) bytes physically (the stream is read twice): the physical rate is "
                          "1.5 N_i / (N_i + N_{i+1}) ~ 1.5 x the algorithmic one"},
        "iteration": {"alg_bytes": b_stats + b_merge, "GBps": round((b_stats + b_merge) / t_iter / 1e9, 1),
                      "frac_of_hbm_peak": frac(b_stats + b_merge, t_iter), "frac_of_measured_copy_peak_6290": round((b_stats + b_merge) / t_iter / 1e9 / 6290.0, 4)},
    }


def run_e2e_class_workload(wl, ref):
    """What a minbpe user calls: RegexTokenizer().train(text, vocab_size) (regex.py:36-70) on the headline input as ONE
    Python str -- wall clock of the whole call: utf-8 encode, native pre-split, (de-duplication), H2D upload, device
    training, merges / vocab dicts.  Both settings of the class's `dedup` switch; the merges must be the headline's."""
    from minbpe_amd import RegexTokenizer
    data = synth_cached(wl["bytes"], wl["seed"])
    text = data.decode("utf-8")
    out = {"workload": f"RegexTokenizer().train(text, {wl['vocab']}) on the {wl['bytes']} B headline input as one str "
                       f"({len(text)} characters): wall clock of the whole call"}
    tok = RegexTokenizer()
    tok.train(text[:1_000_000], 300)  # (context, allocations)
    # ("auto" = off below 2 GiB of text since round 6: tokenizer.py, RegexTokenizer.dedup)
    for label, dedup in (("dedup_auto", "auto"), ("dedup_on", True), ("dedup_off", False)):
        tok = RegexTokenizer()
        tok.dedup = dedup
        t0 = time.perf_counter()
        tok.train(text, wl["vocab"])
        dt = time.perf_counter() - t0
        pairs = [p for p, _ in sorted(tok.merges.items(), key=lambda kv: kv[1])]
        out[label] = {"wall_s": round(dt, 3), "merges_per_s": round((wl["vocab"] - 256) / dt, 1),
                      "same_merges_as_headline_run": None if ref is None else bool(pairs == [tuple(p) for p in ref[0]])}
    out["python_reference"] = ("not timed here (cannot travel to this host); BASELINE.md: 1.07 s per merge at 4 MB in the build "
                               "container, O(N) per merge -> ~268 s per merge at 1 GB, ~98 days for these 31,744 merges")
    return out


def cpu_baseline(wl, data, offs, res, cpu_bytes, cpu_iters, total_bytes=None):
    """The oracle (C port of the reference loop, one thread) on the first `cpu_bytes` of the same input
    for `cpu_iters` iterations; the rate is scaled linearly in N to the full size (the reference loop is
    O(N) per merge).  Also the unmodified Python reference on a smaller slice when it is importable."""
    import numpy as np
    import oracle
    nb = min(cpu_bytes, len(data))
    if offs is not None:
        k = int(np.searchsorted(offs, nb, side="right")) - 1
        nb = int(offs[k]) if nb < len(data) else nb
        so = offs[:k] if nb < len(data) else offs
    else:
        while nb < len(data) and (data[nb] & 0xC0) == 0x80:
            nb -= 1
        so = None
    sample = data[:nb]
    t0 = time.perf_counter()
    cp, _, _ = oracle.train(sample, cpu_iters, so)
    ct = time.perf_counter() - t0
    rate = cpu_iters / ct
    total = total_bytes or len(data)  # (a sharded job: the bytes of ALL ranks)
    scale = nb / total
    c_port = {
        "value": round(rate * scale, 4), "unit": "merges/s", "cores": 1, "kind": "port",
        "sample": f"oracle/bpe_oracle.c (get_stats + max + merge, one thread): first {cpu_iters} merges of the "
                  f"first {nb} bytes of the same input in {ct:.1f} s = {rate:.2f} merges/s, scaled by "
                  f"{scale:.3f} (O(N) per merge) to the full {total} bytes",
        "sample_merges_per_s": round(rate, 3),
    }
    if nb == len(data):
        c_port["gpu_first_merges_equal"] = bool(cp == res["pairs"][:cpu_iters])
    # THE BASELINE (north_star: "minbpe's pure-Python CPU path timed on the GPU box's own host cores in the same run"):
    # oracle/pyref.py restates base.py:13-41 + basic.py:31-42 statement for statement (the reference tree cannot travel
    # to the GPU box; the restatement is pinned against the fixtures the reference generated,
    # tests/test_oracle_golden.py) -- one thread, like the reference.  The C port above rides along as `c_port`.
    out = {"value": None, "unit": "merges/s", "cores": 1, "kind": "port", "sample": None, **host_info(), "c_port": c_port}
    try:
        from oracle import pyref
        py_n = min(PY_SAMPLE_BYTES, nb)
        while py_n < nb and (sample[py_n] & 0xC0) == 0x80:
            py_n -= 1
        py_sample = bytes(sample[:py_n])
        K = PY_SAMPLE_MERGES
        t0 = time.perf_counter()
        pp, _ = pyref.train(py_sample, K)
        pt_total = time.perf_counter() - t0
        pt = pt_total / K
        full = 1.0 / (pt * total / py_n)
        out["value"] = round(full, 6)
        out["sample"] = (f"oracle/pyref.py = minbpe's own loop (get_stats + max + merge, base.py:13-41, basic.py:31-42) in pure "
                         f"Python, one thread, one stream: {K} merges of the first {py_n} bytes of the same input in "
                         f"{pt_total:.1f} s = {pt:.3f} s per merge, scaled linearly in N (the loop is O(N) per merge) to the "
                         f"full {total} bytes")
        out["s_per_merge_on_sample"] = round(pt, 3)
        out["sample_bytes"] = py_n
        out["sample_merges"] = K
        out["merges_per_s_on_sample"] = round(1.0 / pt, 4)
        c_first = oracle.train(py_sample, K)[0] if so is None else None
        if c_first is not None:
            out["equals_c_oracle_first_merges"] = bool([tuple(x) for x in pp] == [tuple(x) for x in c_first])
    except Exception as e:  # the bench line must still come out: the C port is then the figure, labelled
        out.update({k: v for k, v in c_port.items() if k in ("value", "sample")})
        out["python_leg_error"] = f"{type(e).__name__}: {e}"
    return out


ENCODE_DOCS = 1_000_000   # BASELINE.json configs[4] / SURVEY 8d cfg5: a batch of 1M documents
ENCODE_TEXT_BYTES = 760_000_000  # synth_text(seed 4) holds ~1.45 documents per KB: a little over 1M documents


def encode_docs():
    """1,000,000 documents = synth_text (seed 4) cut at its blank lines."""
    import numpy as np
    data = synth_cached(ENCODE_TEXT_BYTES, 4)
    arr = np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero((arr[:-1] == 10) & (arr[1:] == 10)).astype(np.uint64) + 2
    doc_offs = np.unique(np.concatenate([np.zeros(1, np.uint64), nl[nl < len(data)]]))
    if len(doc_offs) > ENCODE_DOCS:
        data = data[:int(doc_offs[ENCODE_DOCS])]
        doc_offs = doc_offs[:ENCODE_DOCS]
    return data, doc_offs


def cfg3_merges(eng):
    """the merges of BASELINE.json configs[2] (the headline training run), trained here when the headline did
    not just produce them"""
    wl = WORKLOADS["regex1g"]
    data, offs, _ = make_input(wl)
    eng.load_bytes(data, offs)
    return eng.train(wl["vocab"] - 256)["pairs"]


def run_encode_workload(eng, steps, warmup, barrier, pairs=None, tables=("cfg3", "cl100k_sized")):
    """configs[4]: batch encode of 1,000,000 documents, GPT-4 split, with (1) the merges of configs[2]
    (vocab 32000, as SURVEY 8d specifies: cl100k ranks are not available offline) and (2) a rank table of
    cl100k_base's size built around them (100,000 merges, non-consecutive ids up to 301,000: tests/helpers.py
    cl100k_shaped_table).  Device-only and PCIe-inclusive rates; the whole batch is compared with oracle.encode."""
    import numpy as np
    from minbpe_amd import _native
    import oracle
    from helpers import cl100k_shaped_table
    if pairs is None:
        pairs = cfg3_merges(eng)
    data, doc_offs = encode_docs()
    offs, _first = _native.split_docs(data, doc_offs, 4)
    n_docs = len(doc_offs)
    out = {}
    for tname in tables:
        if tname == "cfg3":
            tp, mids = np.asarray(pairs, dtype=np.int32), None
            tdesc = f"the {len(tp)} merges of the headline training run (vocab {256 + len(tp)})"
        else:
            tp, mids = cl100k_shaped_table(pairs, 100_000, 9, "sparse")
            tdesc = (f"a rank table of cl100k_base's size: {len(tp)} merges (the headline's {len(pairs)} spread over "
                     f"them, the rest pairs of tokens defined so far), ids 1000 + 3 rank (up to {int(mids.max())})")
        for _ in range(warmup):
            eng.encode_batch(tp, mids, data, offs)
        eng.set_option("profile", 2)
        eng.prof_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ids, out_offs = eng.encode_batch(tp, mids, data, offs)
        barrier()
        dt = (time.perf_counter() - t0) / steps
        prof = eng.prof_read()["encode"]
        eng.set_option("profile", 0)
        dev_s = prof["ms"] * 1e-3 / max(steps, 1)
        # the WHOLE batch against the oracle (oracle.encode does ~70 MB/s on one core), compared by digest of
        # the ids and of the per-chunk output offsets
        t0 = time.perf_counter()
        oid, ooff = oracle.encode(tp, data, offs, merge_ids=mids)
        ct = time.perf_counter() - t0
        equal = bool(len(oid) == len(ids)
                     and hashlib.sha256(np.ascontiguousarray(oid, dtype=np.int32).tobytes()).digest()
                     == hashlib.sha256(np.ascontiguousarray(ids, dtype=np.int32).tobytes()).digest()
                     and np.array_equal(np.asarray(ooff, dtype=np.uint64), np.asarray(out_offs, dtype=np.uint64)))
        alg = int(len(data) + 4 * len(ids))
        ach = alg / dev_s / 1e9 if dev_s else 0.0
        traffic, tsrc = encode_traffic(tname)
        # the same batch with inputs and outputs already in HBM (bpe_encode_batch_resident): what a caller that keeps
        # its batches on the device pays -- wall clock of the call, stream synchronised inside
        resident = None
        try:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            d_bytes = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
            d_offs = torch.from_numpy(np.asarray(offs).astype(np.int64)).to(dev)
            d_ids = torch.empty(len(data), dtype=torch.int32, device=dev)
            d_ooff = torch.empty(len(offs) + 1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            args = (tp, mids, d_bytes.data_ptr(), len(data), d_offs.data_ptr(), len(offs), d_ids.data_ptr(), d_ooff.data_ptr())
            eng.encode_batch_resident(*args)
            t0 = time.perf_counter()
            for _ in range(steps):
                total = eng.encode_batch_resident(*args)
            rdt = (time.perf_counter() - t0) / steps
            same = bool(total == len(ids) and torch.equal(d_ids[:total].cpu(), torch.from_numpy(np.asarray(ids, dtype=np.int32)))
                        and torch.equal(d_ooff.cpu(), torch.from_numpy(np.asarray(out_offs).astype(np.int64))))
            resident = {"ms_per_step_wall": round(rdt * 1e3, 3), "docs_per_s": round(n_docs / rdt, 1),
                        "tokens_per_s": round(total / rdt, 1), "equals_host_form": same}
            del d_bytes, d_offs, d_ids, d_ooff
        except Exception as e:  # (the line must still come out)
            resident = f"failed: {type(e).__name__}: {e}"
        r = {
            "workload": f"batch encode, {n_docs} documents / {len(data)} B synthetic UTF-8 (seed 4) / {len(offs)} "
                        f"GPT-4-split chunks, {tdesc}",
            "docs_per_s_device": round(n_docs / dev_s, 1) if dev_s else None,
            "tokens_per_s_device": round(len(ids) / dev_s, 1) if dev_s else None,
            "text_GBps_device": round(len(data) / dev_s / 1e9, 2) if dev_s else None,
            "docs_per_s_pcie_inclusive": round(n_docs / dt, 1), "tokens_per_s_pcie_inclusive": round(len(ids) / dt, 1),
            "device_resident_batch": resident,
            "ms_per_step": round(dt * 1e3, 2), "device_ms_per_step": round(dev_s * 1e3, 3), "tokens": int(len(ids)),
            "max_token_id": int(ids.max()) if len(ids) else None,
            "parity": {"bytes_checked": len(data), "chunks_checked": int(len(offs)), "tokens_checked": int(len(oid)),
                       "equal_oracle": equal},
            # algorithmic bytes of the encode loop: the text in, 4 bytes per token out
            "alg_bytes_per_step": alg,
            "roofline": {
                "bound": "hbm", "kernel": "bpe_encode_batch on the device: k_enc_pass1 + k_enc_pass2 + k_enc_place_chained "
                                          "(hipEvents around all of them)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                "achieved_kind": "algorithmic (text bytes in + 4 B per token out) / hipEvent time",
                "traffic": traffic, "traffic_source": tsrc,
                "physical_GBps": round(traffic / dev_s / 1e9, 1) if (traffic and dev_s) else None,
                "physical_frac": round(traffic / dev_s / 1e9 / HBM_PEAK_GBPS, 4) if (traffic and dev_s) else None,
                "note": "every distinct chunk is encoded once and copied to its other occurrences (DESIGN 4); what is left "
                        "is one random 32-byte table access per chunk in each pass (its slot, then its owner's tokens) "
                        "-- sectors served by L2 / Infinity Cache, not counted in the algorithmic bytes -- and pass 1's "
                        "instruction issue"},
            "cpu_baseline": {"value": round(len(data) / ct, 1), "unit": "bytes/s", "cores": 1, "kind": "port",
                             "sample": f"oracle.encode on the whole batch ({len(data)} bytes in {ct:.1f} s)", **host_info()},
        }
        del ids, out_offs, oid, ooff
        out[tname] = r
    first = out[tables[0]]
    for tname in tables[1:]:
        first[tname] = out[tname]
    return first


def run_encode_long_workload(eng, pairs=None, n_docs=100_000):
    """The encode paths the 1 M-document batch does not reach (its synthetic prose holds no chunk above 32 bytes):
    (1) `long_chunks`: n_docs documents of the same text with URLs, identifiers, whitespace runs and letter noise spliced
    in -- at least 1 % of the GPT-4-split chunks are 33 .. 4096 bytes (one or four waves per chunk with the chunk in LDS,
    k_enc_long) and a few are longer (up to 9001 bytes: the sixteen-wave instantiation, up to 9216; beyond that the
    stream-wide rounds) -- the whole batch against oracle.encode;
    (2) `basic_100mb`: BasicTokenizer.encode (basic.py:57-74: ONE chunk) of the 100 MB of configs[1] with its own 3840
    merges.  oracle.encode is O(N x rounds) on one chunk (hours at this size): the answer is checked in full against the
    stream TRAINING leaves resident for the same text (training applies the merges in order, which is encode of its own
    input -- and that stream is pinned to the oracle: its length after every merge is in the golden digests of cfg2), and
    against oracle.encode itself on a 300 kB text."""
    import numpy as np
    from minbpe_amd import _native
    import oracle
    out = {}
    if pairs is None:
        pairs = cfg3_merges(eng)
    tp = np.asarray(pairs, dtype=np.int32)
    # ---- (1) documents with long chunks
    data, doc_offs = encode_docs()
    doc_offs = doc_offs[:n_docs + 1]
    text = bytes(data[:int(doc_offs[-1])])
    rng = np.random.default_rng(7)
    letters = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz", dtype=np.uint8)
    urlc = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz/._-", dtype=np.uint8)
    parts = []

    def long_item(k):
        if k < 5:
            return b" https://" + rng.choice(urlc, int(rng.integers(30, 200))).tobytes()
        if k < 8:
            return b" " + rng.choice(letters, int(rng.integers(33, 400))).tobytes()
        return b" " * int(rng.integers(34, 300))

    for d in range(len(doc_offs) - 1):
        a, b = int(doc_offs[d]), int(doc_offs[d + 1])
        cuts = []
        for f in (1, 2):
            cut = a + (b - a) * f // 3
            while cut < b and text[cut] != 32:
                cut += 1
            cuts.append(cut)
        parts.append(text[a:cuts[0]])
        parts.append(long_item(d % 10))
        parts.append(text[cuts[0]:cuts[1]])
        parts.append(long_item((d // 3) % 10))
        if d % 2 == 0:
            parts.append(long_item(5 + (d // 7) % 5))
        if d % 2000 == 7:
            parts.append(b" " + rng.choice(letters, int(rng.integers(1000, 9000))).tobytes())
        parts.append(text[cuts[1]:b])
    text2 = b"".join(parts)
    arr = np.frombuffer(text2, dtype=np.uint8)
    nl = np.flatnonzero((arr[:-1] == 10) & (arr[1:] == 10)).astype(np.uint64) + 2
    doc2 = np.unique(np.concatenate([np.zeros(1, np.uint64), nl[nl < len(text2)]]))
    offs, _first = _native.split_docs(text2, doc2, 4)
    lens = np.diff(np.append(np.asarray(offs, dtype=np.uint64), np.uint64(len(text2)))).astype(np.int64)
    eng.encode_batch(tp, None, text2, offs)
    eng.set_option("profile", 2)
    eng.prof_reset()
    t0 = time.perf_counter()
    ids, out_offs = eng.encode_batch(tp, None, text2, offs)
    dt = time.perf_counter() - t0
    dev_ms = eng.prof_read()["encode"]["ms"]
    eng.set_option("profile", 0)
    t0 = time.perf_counter()
    oid, ooff = oracle.encode(tp, text2, offs)
    ct = time.perf_counter() - t0
    out["long_chunks"] = {
        "workload": f"{len(doc2)} documents / {len(text2)} B / {len(offs)} GPT-4-split chunks, the headline's {len(tp)} merges",
        "chunks_33_to_4096_B": int(((lens > 32) & (lens <= 4096)).sum()), "chunks_over_4096_B": int((lens > 4096).sum()),
        "share_of_chunks_over_32_B": round(float((lens > 32).mean()), 4),
        "share_of_bytes_in_chunks_over_32_B": round(float(lens[lens > 32].sum() / max(len(text2), 1)), 4),
        "device_ms": round(dev_ms, 3), "ms_wall": round(dt * 1e3, 2), "tokens": int(len(ids)),
        "parity": {"tokens_checked": int(len(oid)), "equal_oracle": bool(
            len(oid) == len(ids) and np.array_equal(np.asarray(oid, dtype=np.int32), np.asarray(ids, dtype=np.int32))
            and np.array_equal(np.asarray(ooff, dtype=np.uint64), np.asarray(out_offs, dtype=np.uint64)))},
        "cpu_oracle_s": round(ct, 2),
    }
    del ids, out_offs, oid, ooff
    # ---- (2) BasicTokenizer.encode of 100 MB
    wl = dict(WORKLOADS["cfg2"])
    bdata, _o, _ = make_input(wl)
    nm = wl["vocab"] - 256
    eng.load_bytes(bdata)
    bp = np.asarray(eng.train(nm)["pairs"], dtype=np.int32)
    trained = eng.read_ids().copy()
    t0 = time.perf_counter()
    bids, _boff = eng.encode_batch(bp, None, bdata, None)
    bdt = time.perf_counter() - t0
    small = bytes(synth_cached(100_000_000, wl["seed"])[50_000_000:50_300_000])
    while small and (small[0] & 0xC0) == 0x80:
        small = small[1:]
    sid, _so = eng.encode_batch(bp, None, small, None)
    oid, _oo = oracle.encode(bp, small, None)
    out["basic_100mb"] = {
        "workload": f"BasicTokenizer.encode: {len(bdata)} B as ONE chunk, its own {nm} merges (configs[1])",
        "s_wall": round(bdt, 3), "MB_per_s": round(len(bdata) / bdt / 1e6, 1), "tokens": int(len(bids)),
        "parity": {"equals_the_stream_training_leaves": bool(np.array_equal(bids, trained)),
                   "equal_oracle_on_300kB": bool(np.array_equal(np.asarray(oid, dtype=np.int32), sid))},
    }
    return out


def encode_traffic(tname):
    """HBM bytes of one encode step from the committed PMC pass of this command (profiles/), attached only
    when it was measured on these library sources."""
    f = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_encode_pmc.json")
    if not os.path.exists(f):
        return None, None
    with open(f) as fh:
        pmc = json.load(fh)
    if pmc.get("source_hash") != source_hash() or tname not in pmc.get("hbm_bytes_per_step", {}):
        return None, "committed PMC profile is from other library sources: not attached"
    return int(pmc["hbm_bytes_per_step"][tname]), f"profiles/{PROFILE_ROUND}_encode_pmc.json (source_hash {pmc['source_hash']})"


def parity_failures(obj, path=""):
    """Every place in the line where a comparison with the oracle / the plain run came out False (None = not checked)."""
    bad = []
    keys = ("equal", "equal_oracle", "equals_single_gpu", "ranks_agree", "same_merges_and_counts_as_plain_run",
            "same_merges_as_headline_run", "same_merges_and_counts_as_headline_run", "equals_the_stream_training_leaves", "equal_oracle_on_300kB", "equals_host_form",
            "len_drop_equals_count_and_counts_monotone", "split_equals_regex_module", "parity_equal", "invariants_hold")
    if isinstance(obj, dict):
        for k, v in obj.items():
            if k in keys and v is False:
                bad.append(f"{path}{k}")
            elif isinstance(v, (dict, list)):
                bad += parity_failures(v, f"{path}{k}.")
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            bad += parity_failures(v, f"{path}{i}.")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, help="regex1g (default) | basic1g | cfg2 | encode")
    ap.add_argument("--secondary", default=None,
                    help="comma list of further workloads reported under 'secondary' (default at N=1 with the "
                         "default headline: basic1g,cfg2; 'none' to skip)")
    ap.add_argument("--secondary-steps", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=None, help="override the workload's stream size (per GPU)")
    ap.add_argument("--vocab", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--mode", type=int, default=int(os.environ.get("BPE_MODE", "-1")),
                    help="-1 library default | 0 recount (the literal get_stats-every-iteration loop) | 1 delta")
    ap.add_argument("--cpu-iters", type=int, default=40, help="oracle iterations for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-bytes", type=int, default=100_000_000, help="slice of the input the oracle is timed on")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (experiments)")
    args = ap.parse_args()

    import torch  # device sync + torch.distributed (RCCL) plumbing only
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    force_dp = os.environ.get("BENCH_FORCE_DP") == "1"  # exercise the sharded path on one GPU
    sharded = world > 1 or force_dp
    if sharded:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import minbpe_amd
    from minbpe_amd import Engine

    name = args.workload or "regex1g"
    eng = Engine(local_rank)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(dt):
        if not sharded:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    line = {"metric": "BPE merges/sec", "unit": "merges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic", "library": minbpe_amd.version(), "source_hash": source_hash()}

    if name == "encode_long":
        line.update({"metric": "batch encode, long chunks", "unit": "ms", "value": None, "encode_long": run_encode_long_workload(eng)})
        print(json.dumps(line))
        eng.close()
        return
    if name == "encode":
        r = run_encode_workload(eng, args.steps, args.warmup, barrier)
        line.update({"metric": "batch encode docs/sec", "unit": "docs/s", "value": r["docs_per_s_device"],
                     "ms_per_step": r["device_ms_per_step"], "config": {"workload": r["workload"]},
                     "roofline": r.pop("roofline"), "cpu_baseline": r.pop("cpu_baseline"), "encode": r})
        print(json.dumps(line))
        eng.close()
        return

    wl = dict(WORKLOADS[name])
    if wl.get("dedup"):
        raise SystemExit(f"{name} is a secondary figure: python bench.py --secondary {name}")
    for k in ("bytes", "vocab", "seed"):
        if getattr(args, k) is not None:
            wl[k] = getattr(args, k)

    if not sharded:
        r, data, offs, res = run_train_workload(name, wl, eng, args.steps, args.warmup, barrier, reduce_max, args.mode)
        cpu = None
        if args.cpu_iters > 0:
            cpu = cpu_baseline(wl, data, offs, res, args.cpu_bytes, args.cpu_iters)
        par = r["parity"]
        inv_ok = r["invariants"]["len_drop_equals_count_and_counts_monotone"]
        nm_ = wl["vocab"] - 256
        # (the driver keeps the first 120 characters of config.workload: the parity verdict comes first)
        short = (f"parity {par['merges_checked']}/{nm_} merges = oracle: {par['equal']}" if par["golden"]
                 else "parity: no committed oracle digest") + \
            f"; {name}: {'Regex' if wl['chunked'] else 'Basic'}Tokenizer.train, {wl['bytes'] / 1e9:g} GB synth seed {wl['seed']}, vocab {wl['vocab']}"
        headline_ok = (par["equal"] is not False) and inv_ok
        line.update({
            # a headline whose merges differ from the oracle's is not a measurement
            "value": r["merges_per_s"] if headline_ok else None, "ms_per_step": r["ms_per_step"],
            "config": {"workload": short[:119],
                       "parity_equal": par["equal"], "merges_checked": par["merges_checked"], "invariants_hold": inv_ok,
                       "workload_detail": r["workload"] + "; " + (
                           f"first {par['merges_checked']} merges equal the oracle's committed digests: {par['equal']}"
                           if par["golden"] else "no committed oracle digest for this input") +
                       f"; full-length invariants hold: {inv_ok}",
                       "mode": "recount" if args.mode == 0 else ("delta" if args.mode == 1 else "default"),
                       "parallelism": "single"},
            "roofline": r.pop("roofline"), "cpu_baseline": cpu,
        })
        line.update({k: v for k, v in r.items() if k not in ("workload", "merges_per_s", "ms_per_step", "steps")})
        plain_ref = (res["pairs"], res["counts"]) if (name == "regex1g" and wl == WORKLOADS["regex1g"]) else None
        del data, offs, res
        sec = args.secondary
        if sec is None:
            sec = ("basic1g,cfg2,regex1g_dedup,e2e_class,literal_path,encode,encode_long"
                   if (args.workload is None and args.bytes is None and args.vocab is None) else "none")
        secondary = {}
        for sname in [s for s in sec.split(",") if s and s != "none"]:
            try:
                if sname == "encode":  # BASELINE.json configs[4] (see run_encode_workload)
                    secondary[sname] = run_encode_workload(eng, 2, 1, barrier, plain_ref[0] if plain_ref else None)
                    continue
                if sname == "encode_long":  # long chunks and BasicTokenizer.encode (see run_encode_long_workload)
                    secondary[sname] = run_encode_long_workload(eng, plain_ref[0] if plain_ref else None)
                    continue
                if sname == "e2e_class":
                    secondary[sname] = run_e2e_class_workload(dict(WORKLOADS["regex1g"]), plain_ref)
                    continue
                if sname == "literal_path":  # the loop as the reference writes it: get_stats + merge of every iteration
                    secondary[sname] = run_literal_path_workload(dict(WORKLOADS["regex1g"]), eng, plain_ref)
                    continue
                if WORKLOADS[sname].get("dedup"):
                    secondary[sname] = run_dedup_workload(dict(WORKLOADS[sname]), eng, args.secondary_steps,
                                                          barrier, plain_ref)
                    continue
                sr, _d, _o, _r = run_train_workload(sname, dict(WORKLOADS[sname]), eng, args.secondary_steps, 0,
                                                    barrier, reduce_max, args.mode, secondary=True)
                del _d, _o, _r
                secondary[sname] = sr
            except Exception as e:  # the headline line must still come out
                secondary[sname] = f"failed: {type(e).__name__}: {e}"
        line["secondary"] = secondary
    else:
        # BasicTokenizer's single stream does not shard; N > 1 runs the chunked
        # (RegexTokenizer) training sharded by chunks: every rank generates and splits its own
        # `bytes` of text (weak scaling, cfg4 shape), the two per-merge all-reduces go over RCCL.
        from minbpe_amd.dist import GpuShard, TorchComm, train_sharded, init_native_comm
        import numpy as np
        num_merges = wl["vocab"] - 256
        data, offs, _ = make_input(wl, rank)
        eng.load_bytes(data, offs)
        comm = TorchComm()
        # BPE_DIST = native (default): bpe_dp_train, chain steps with the library's own librccl | torch: the same loop,
        # every collective a torch.distributed all-reduce through a callback | steps: the per-merge protocol of dist.py
        which = os.environ.get("BPE_DIST", "native")
        if which == "native" and init_native_comm(eng, comm):
            dist_path = "chain steps, librccl in the library's loop (bpe_dp_train)"
            step = lambda: eng.dp_train(num_merges)
        elif which != "steps":
            from minbpe_amd.dist import torch_allreduce
            dev = torch.device("cuda", local_rank)
            eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            ar = torch_allreduce(comm, dev)
            dist_path = "chain steps, torch.distributed all-reduce per collective (bpe_dp_train_cb)"
            step = lambda: eng.dp_train_cb(num_merges, rank, world, ar)
        else:
            shard = GpuShard(eng, local_rank)
            dist_path = "per-merge protocol, torch.distributed (dist.train_sharded)"
            step = lambda: train_sharded(shard, comm, num_merges)
        for _ in range(args.warmup):
            step()
        try:  # hipEvents around this rank's merge passes during the timed steps (as at N = 1)
            eng.set_option("profile", 1)
            eng.prof_reset()
        except Exception:
            pass
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        barrier()
        dt = reduce_max(time.perf_counter() - t0)
        roofline = None
        try:
            mg = eng.prof_read()["merge"]
            eng.set_option("profile", 0)
            n0 = torch.tensor([len(data)], dtype=torch.int64, device="cuda")
            dist.all_reduce(n0, op=dist.ReduceOp.SUM)
            lens = [int(n0.item())] + [int(x) for x in res["lens"]]  # GLOBAL stream lengths
            alg = sum(4 * (2 * lens[i] + lens[i + 1]) for i in range(len(lens) - 1)) * args.steps / world
            if mg["ms"] > 0 and mg["launches"]:
                ach = alg / (mg["ms"] * 1e-3) / 1e9
                roofline = {
                    "bound": "hbm", "kernel": "merge pass of rank 0 on its shard: k_merge_chain | k_merge_chain_dense | "
                                             "k_merge_chain_dense1 (a chain step) | k_merge_ab_* + k_merge_aa (a general iteration)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                    "launches": mg["launches"], "avg_launch_ms": round(mg["ms"] / mg["launches"], 5),
                    "alg_bytes_per_launch": int(alg // mg["launches"]),
                    "achieved_kind": "algorithmic (SURVEY 8d bytes / hipEvent time)",
                    "note": "per GPU: the job's algorithmic bytes (4(2N_i + N_{i+1}) on GLOBAL lengths) / world, over "
                            "the hipEvent time of rank 0's merge passes",
                }
                # physical: counters cannot be read inside a run; a rank's merge passes work on a shard of the single-GPU
                # headline's size with the same kernels, so the committed single-GPU PMC profile's bytes per merge give
                # an ESTIMATE of this rank's traffic (labelled as such; the per-launch figure does not carry over:
                # a sharded step's batch is capped lower)
                pmc_file = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_regex1g_pmc.json")
                if os.path.exists(pmc_file) and wl["bytes"] == WORKLOADS["regex1g"]["bytes"]:
                    with open(pmc_file) as f:
                        pmc = json.load(f)
                    if pmc.get("source_hash") == source_hash() and pmc.get("merges"):
                        per_merge = pmc["hbm_bytes_total"] / pmc["merges"]
                        phys = per_merge * num_merges * args.steps / (mg["ms"] * 1e-3) / 1e9
                        roofline.update({
                            "achieved": round(phys, 1), "frac": round(phys / HBM_PEAK_GBPS, 4),
                            "traffic": int(per_merge * num_merges / max(mg["launches"] / args.steps, 1)),
                            "achieved_kind": "physical ESTIMATE: HBM bytes per merge of the single-GPU PMC profile "
                                             f"(profiles/{PROFILE_ROUND}_regex1g_pmc.json, same sources, same shard size) "
                                             "x merges / this rank's hipEvent time of its merge passes",
                            "equivalent_work_GBps": round(ach, 1)})
        except Exception as e:
            roofline = None
            line["roofline_error"] = f"{type(e).__name__}: {e}"
        digest = int.from_bytes(hashlib.sha256(repr((res["pairs"], res["counts"], res["lens"])).encode())
                                .digest()[:7], "big")
        lo = torch.tensor([digest], dtype=torch.int64, device="cuda")
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dp_check = {"ranks_agree": bool(lo.item() == hi.item()), "equals_single_gpu": None}
        # the job's merges against the CPU oracle's: tests/golden/big_golden.json[regex1g_dp<world>_w] = the weighted
        # oracle on the distinct chunks of the shards back to back (all 31,744 merges, pairs + counts + GLOBAL lengths);
        # every rank vouches for its own shard's bytes (sha256), rank 0 compares the digests
        g, mine = None, False
        try:
            g = golden_entry(f"{name}_dp{world}_w")
            if world == 1:  # (BENCH_FORCE_DP: one shard = the single-GPU headline input)
                g1 = golden_entry(f"{name}_w")
                g = dict(g1, world=1, shard_sha256=[g1["data_sha256"]]) if g1 else None
            if not (g and g.get("world") == world and g["bytes"] == wl["bytes"] and g["seed"] == wl["seed"]):
                g = None
            mine = bool(g) and hashlib.sha256(data).hexdigest() == g["shard_sha256"][rank]
        except Exception:  # (a malformed entry: nothing is claimed; every rank still joins the reduction below)
            g, mine = None, False
        ok = torch.tensor([1 if mine else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if g is None:
            dp_check["oracle"] = f"no committed oracle digest for {world} shard(s) of this size"
        elif not bool(ok.item()):
            dp_check["oracle"] = "not checked: a shard's bytes differ from the golden's"
        elif rank == 0:
            try:
                from helpers import checkpoint_digests, first_divergence
                k = min(g["done"], len(res["pairs"]))
                got = checkpoint_digests(res["pairs"][:k], res["counts"][:k], res["lens"][:k], g["step"])
                bad = first_divergence(got, g["digests"])
                checked = [c for c, _ in got if c in {c2 for c2, _ in g["digests"]}]
                dp_check["oracle"] = {"entry": f"tests/golden/big_golden.json[{name}_dp{world}_w] (weighted oracle on the "
                                               f"distinct chunks of the {world} shard(s) back to back, {g['done']} merges)",
                                      "merges_checked": max(checked) if checked else 0,
                                      "equal": bool(checked) and bad is None,
                                      **({"first_bad_checkpoint": bad} if bad is not None else {})}
            except Exception as e:
                dp_check["oracle"] = f"not checked: {type(e).__name__}: {e}"
        if rank == 0 and os.environ.get("BENCH_DP_CHECK", "1") == "1" and wl["bytes"] * world <= 2_000_000_000:
            try:  # the sharded result must be the single-GPU result on the concatenation of all shards
                parts, offl, base = [], [], 0
                for r_ in range(world):
                    d, o, _ = make_input(wl, r_)
                    parts.append(d)
                    offl.append(o + np.uint64(base))
                    base += len(d)
                eng2 = Engine(local_rank)
                eng2.load_bytes(b"".join(parts), np.concatenate(offl))
                single = eng2.train(num_merges)
                eng2.close()
                dp_check["equals_single_gpu"] = bool(
                    single["pairs"] == res["pairs"] and single["counts"] == res["counts"]
                    and single["lens"] == res["lens"])
            except Exception as e:
                dp_check["equals_single_gpu"] = f"not checked: {type(e).__name__}: {e}"
        line.update({
            # the units ALL ranks processed / the time (the bench contract's whole-job aggregate under weak scaling): a
            # merge of the job is applied to every rank's 1 GB shard, so N ranks do N x num_merges shard-merges per train;
            # the job's own rate (what a user waits for) is job_merges_per_s
            "value": round(world * num_merges * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 3),
            # (N > 1: `value` counts a merge once per shard it is applied to -- the unit says so, so that nobody reads an
            # N-fold rise of the job's own rate into it; at N = 1 the two are one and the unit is plain merges/s)
            "unit": "merges/s" if world == 1 else "shard-merges/s",
            "job_merges_per_s": round(num_merges * args.steps / dt, 2),
            "config": {"workload": f"{wl['desc']} sharded over {world} GPUs by contiguous chunk ranges, "
                                   f"{wl['bytes']} B synthetic UTF-8 per GPU (seed {wl['seed']}+rank), vocab "
                                   f"{wl['vocab']} ({num_merges} merges); ranks agree: {dp_check['ranks_agree']}",
                       "parallelism": f"dp{world} ({dist_path}: per step of 1..K merges one MIN all-reduce of a tie's "
                                      f"first occurrences and one SUM all-reduce of the batch's table deltas)"},
            "value_definition": f"shard-merges per second = {world} ranks x the job's merges per second: every merge of the "
                                f"ONE sharded job is applied to each rank's {wl['bytes']} B shard (weak scaling: the work per "
                                "rank is fixed, the job grows with N); job_merges_per_s = the job's own rate",
            "sharded_check": dp_check, "roofline": roofline, "cpu_baseline": None,
            "merge_passes": eng.train_stats(),
        })
        if rank == 0 and args.cpu_iters > 0:  # (rank 0's host, its own shard as the sample, scaled to the whole job)
            try:
                line["cpu_baseline"] = cpu_baseline(wl, data, offs, res, args.cpu_bytes, args.cpu_iters,
                                                    total_bytes=wl["bytes"] * world)
                # (its `value` is the CPU's rate on the WHOLE job in the job's merges per second: set it against
                # job_merges_per_s, or times N against `value`)
                line["cpu_baseline"]["compare_with"] = "job_merges_per_s"
            except Exception as e:
                line["cpu_baseline"] = f"failed: {type(e).__name__}: {e}"

    bad = parity_failures(line)
    if bad:
        line["parity_failures"] = bad
    if rank == 0:
        print(json.dumps(line))
    eng.close()
    if sharded:
        dist.destroy_process_group()
    if bad:  # a fast run whose results differ from the reference's is not done: say so to whoever looks at the exit code
        print("bench.py: PARITY FAILED: " + "; ".join(bad), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
