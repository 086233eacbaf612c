#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 300 python bench.py --steps 1 --warmup 0 --cpu-iters 0 --secondary regex1g_dedup > gpurun_out/bench_dedup.json 2> gpurun_out/bench_dedup.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_dedup.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac_physical"])
print(d["secondary"])
PY
grep -v "amdgpu.ids" gpurun_out/bench_dedup.err | tail -5 | cut -c1-300
