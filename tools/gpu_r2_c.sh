#!/bin/bash
# Round 2, call C: pair-Bloom slot index, spread removal counters.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-12} -k "not cfg2_all and not cfg3_shape ${PYTEST_K}" > gpurun_out/pytest_gpu.log 2>&1
RC=$?; echo "pytest rc=$RC"; tail -30 gpurun_out/pytest_gpu.log | cut -c1-300
if [ $RC -ne 0 ]; then echo "tests failed: stopping here"; exit 1; fi
timeout -k 5 300 python tools/iter_profile.py regex1g ${ITER_OPTS} > gpurun_out/iter_regex1g_c.json 2> gpurun_out/iter_regex1g_c.err; echo "iter rc=$?"; tail -12 gpurun_out/iter_regex1g_c.err | cut -c1-250
python -c "
import json; d=json.load(open('gpurun_out/iter_regex1g_c.json')); print(d['passes'], d['total_ms'], d['device_ms_by_class'])"
rm -rf gpurun_out/prof_c
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c -o run -- python bench.py --steps 1 --warmup 0 --secondary none --cpu-iters 0 > gpurun_out/prof_c.log 2>&1
echo "prof rc=$?"
DB=$(find gpurun_out/prof_c -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/prof_c_kernel_stats.csv; rm -rf gpurun_out/prof_c; fi
head -12 gpurun_out/prof_c_kernel_stats.csv | cut -c1-60,200-400
tail -2 gpurun_out/prof_c.log | cut -c1-600
