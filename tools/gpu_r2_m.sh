#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
timeout -k 5 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_m.json 2> gpurun_out/bench_m.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_m.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "device_ms_per_step", "parity", "invariants", "merge_passes")})
    print(d["config"]["workload"])
    print(d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"], d["roofline"]["frac_physical"])
    print(d["cpu_baseline"])
    for k, v in d["secondary"].items():
        print(k, v if isinstance(v, str) else {x: v[x] for x in ("merges_per_s", "ms_per_step", "device_ms_per_step", "parity", "invariants", "merge_passes")})
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 gpurun_out/bench_m.err
timeout -k 5 600 python bench.py --workload encode --steps 2 --warmup 1 > gpurun_out/bench_encode.json 2> gpurun_out/bench_encode.err; echo "encode rc=$?"; cut -c1-1500 gpurun_out/bench_encode.json; tail -3 gpurun_out/bench_encode.err
timeout -k 5 300 python tools/microbench_k1.py > gpurun_out/microbench_k1.log 2>&1; echo "k1 rc=$?"; grep -v amdgpu gpurun_out/microbench_k1.log | head -12 | cut -c1-250
