#!/bin/bash
# Round 2, final validation, part 2: the bench lines (default = headline + secondaries + cpu
# baseline; encode), after profiles/r2_regex1g_pmc.json of the same sources was committed.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 900 python bench.py ${BENCH_ARGS} > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_final.json"))
    print({k: d.get(k) for k in ("value", "ms_per_step", "device_ms_per_step", "parity", "invariants", "merge_passes", "source_hash")})
    print(d["config"]["workload"])
    print(d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"], d["roofline"]["frac_physical"])
    print(d["cpu_baseline"])
    for k, v in d.get("secondary", {}).items():
        print(k, v if isinstance(v, str) else {x: v.get(x) for x in ("merges_per_s", "ms_per_step", "device_ms_per_step", "parity", "invariants", "merge_passes")})
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 gpurun_out/bench_final.err
timeout -k 5 600 python bench.py --workload encode --steps 3 --warmup 1 > gpurun_out/bench_encode.json 2> gpurun_out/bench_encode.err; echo "encode rc=$?"; cut -c1-1800 gpurun_out/bench_encode.json; tail -3 gpurun_out/bench_encode.err
