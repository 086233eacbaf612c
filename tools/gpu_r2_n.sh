#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
timeout -k 5 400 python tools/early_profile.py > gpurun_out/early_profile.log 2>&1; echo "early rc=$?"; grep -v amdgpu gpurun_out/early_profile.log | cut -c1-300
