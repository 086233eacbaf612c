#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 600 python -X faulthandler -m pytest tests/test_gpu_big.py -m gpu -q -k "not cfg3_shape" > gpurun_out/pytest_big.log 2>&1
echo "pytest big rc=$?"; tail -6 gpurun_out/pytest_big.log | cut -c1-300
timeout -k 5 300 python tools/iter_profile.py regex1g sparse_ratio=1 > gpurun_out/iter_regex1g_g1.json 2> gpurun_out/iter_regex1g_g1.err; echo "iter ratio 1 rc=$?"; tail -10 gpurun_out/iter_regex1g_g1.err | cut -c1-120
python -c "
import json; d=json.load(open('gpurun_out/iter_regex1g_g1.json')); print(d['passes'], d['total_ms'], d['device_ms_by_class'])"
bash tools/gpu_pmc.sh regex1g
