#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 600 python -X faulthandler -m pytest tests -m gpu -x -q -k "not cfg3_shape" > gpurun_out/dbg.log 2>&1
RC=$?; echo "pytest rc=$RC"; tail -3 gpurun_out/dbg.log | cut -c1-300
if [ $RC -ne 0 ]; then grep -v "site-packages\|dist-packages" gpurun_out/dbg.log | tail -60 | cut -c1-300; exit 1; fi
timeout -k 5 300 python tools/iter_profile.py regex1g ${ITER_OPTS} > gpurun_out/iter_regex1g_u.json 2> gpurun_out/iter_regex1g_u.err; echo "iter rc=$?"; grep -v amdgpu.ids gpurun_out/iter_regex1g_u.err | tail -12 | cut -c1-250
python -c "
import json; d=json.load(open('gpurun_out/iter_regex1g_u.json')); print(d['passes'], d['total_ms'], d['device_ms_by_class'])"
