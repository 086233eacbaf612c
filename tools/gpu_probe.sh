#!/bin/bash
# Which load order of the two HIP runtimes (ours via /opt/rocm, torch's private copy) works?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
K1="dp_train_sharded_solo"
K2="classes_golden or dp_train_sharded_solo or dp_native_rccl_solo"
MINBPE_TEST_TORCH_LATE=1 timeout -k 5 25 python -m pytest tests -m gpu -q -k "$K1" > gpurun_out/probe_a.log 2>&1; echo "A (lib first, torch late) rc=$?"; tail -1 gpurun_out/probe_a.log
MINBPE_TEST_TORCH_LATE=1 timeout -k 5 30 python -m pytest tests -m gpu -q -k "$K2" > gpurun_out/probe_b.log 2>&1; echo "B (dedup, then torch late) rc=$?"; tail -1 gpurun_out/probe_b.log
timeout -k 5 30 python -m pytest tests -m gpu -q -k "$K2 or weighted_sharded" > gpurun_out/probe_c.log 2>&1; echo "C (torch first) rc=$?"; tail -1 gpurun_out/probe_c.log
