#!/bin/bash
# Last GPU call of a round: all GPU tests (no -x: every failure is listed), then the
# de-duplicated vs plain chunked training comparison.  Every step is bounded; nothing reads stdin.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 ${TEST_TIMEOUT:-90} python -X faulthandler -m pytest tests -m gpu -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
timeout -k 5 25 python tools/run_dedup.py > gpurun_out/dedup.log 2>&1; echo "dedup rc=$?"; tail -2 gpurun_out/dedup.log
