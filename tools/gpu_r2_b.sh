#!/bin/bash
# Round 2, call B: first run of the second slotted form (in-place a != b passes, inverted index).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-12} -k "not cfg2_all and not cfg3_shape ${PYTEST_K}" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log | cut -c1-300
timeout -k 5 300 python tools/iter_profile.py regex1g > gpurun_out/iter_regex1g_b.json 2> gpurun_out/iter_regex1g_b.err; echo "iter rc=$?"; tail -12 gpurun_out/iter_regex1g_b.err | cut -c1-250
timeout -k 5 300 python tools/iter_profile.py regex1g sparse=0 > gpurun_out/iter_regex1g_b_dense.json 2> gpurun_out/iter_regex1g_b_dense.err; echo "iter dense rc=$?"; tail -11 gpurun_out/iter_regex1g_b_dense.err | cut -c1-250
timeout -k 5 600 python bench.py --steps 1 --warmup 0 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_b.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "device_ms_per_step", "parity", "invariants")})
    print(d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
    for k, v in d["secondary"].items():
        print(k, v if isinstance(v, str) else {x: v[x] for x in ("merges_per_s", "ms_per_step", "device_ms_per_step", "invariants")})
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 gpurun_out/bench_b.err
