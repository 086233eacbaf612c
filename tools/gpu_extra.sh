#!/bin/bash
# One gpurun call after tools/gpu_final.sh on the same sources: the PMC passes of the secondary workloads (bench.py
# attaches profiles/${TAG}_<workload>_pmc.json by source hash) and the 3.9 GB single-GPU run.
#   gpurun --timeout 900 -- 'TAG=r4 bash tools/gpu_extra.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
TAG=${TAG:-r4}
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for wl in ${PMC_WL:-cfg2 basic1g}; do
    SKIP_KT=1 TAG=$TAG bash tools/gpu_pmc.sh $wl > gpurun_out/${TAG}_pmc_$wl.log 2>&1; el "pmc $wl done"; tail -5 gpurun_out/${TAG}_pmc_$wl.log | cut -c1-200
done
if [ -z "$SKIP_BIG" ]; then
timeout -k 5 300 python bench.py --bytes 3900000000 --steps 1 --warmup 0 --secondary none --cpu-iters 0 > gpurun_out/${TAG}_big_3p9gb_bench.json 2> gpurun_out/${TAG}_big.err
el "big rc=$?"; cut -c1-500 gpurun_out/${TAG}_big_3p9gb_bench.json; tail -2 gpurun_out/${TAG}_big.err
fi
el "done"
