#!/bin/bash
# quick check of the chain-step engine: parity subset, per-merge profile, per-kernel phases (TAG = output prefix)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
TAG=${TAG:-r4_x}
(timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -m gpu -x -q -k "train_synth_2mb or (full8r and (1-1 or 1-7))") > gpurun_out/${TAG}_pytest.log 2>&1; tail -2 gpurun_out/${TAG}_pytest.log
timeout 280 python tools/iter_profile.py regex1g $OPTS > gpurun_out/${TAG}_iter.json 2> gpurun_out/${TAG}_iter.err; tail -10 gpurun_out/${TAG}_iter.err | cut -c1-110
rm -rf gpurun_out/prof_kt
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o run -- python tools/train_n.py regex1g 31744 $OPTS > gpurun_out/prof_kt.log 2>&1; echo rc=$?
DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python tools/rocpd_stats.py "$DB" > gpurun_out/${TAG}_kernel_stats_one_train.csv
python tools/rocpd_phases.py "$DB" 0 100 300 1000 2000 4000 6000 8000 10000 1073741824 > gpurun_out/${TAG}_phases.json
rm -rf gpurun_out/prof_kt
head -8 gpurun_out/${TAG}_kernel_stats_one_train.csv | cut -c1-60,100-190
