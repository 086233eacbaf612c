#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=12 -k "not cfg2_all and not cfg3_shape" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log | cut -c1-300
for R in 3 2 5; do
timeout -k 5 300 python tools/iter_profile.py regex1g sparse_ratio=$R > gpurun_out/iter_regex1g_e$R.json 2> gpurun_out/iter_regex1g_e$R.err; echo "iter ratio $R rc=$?"; tail -10 gpurun_out/iter_regex1g_e$R.err | cut -c1-120
python -c "
import json; d=json.load(open('gpurun_out/iter_regex1g_e$R.json')); print(d['passes'], d['total_ms'], d['device_ms_by_class'])"
done
