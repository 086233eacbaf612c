#!/bin/bash
# One gpurun call = the round's evidence set on the committed sources: GPU parity tests, smoke, the default bench
# line, kernel-trace stats + the PMC pass of the headline (tools/gpu_pmc.sh), the PMC pass of the encode leg.
#   gpurun --timeout 1100 -- 'TAG=r4 bash tools/gpu_final.sh'
# Every step is bounded by `timeout -k`; nothing reads stdin; outputs under gpurun_out/${TAG}_*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
TAG=${TAG:-r4}
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
if [ -z "$SKIP_TESTS" ]; then
timeout -k 5 ${TEST_TIMEOUT:-420} python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1
el "pytest rc=$?"; tail -4 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
timeout -k 5 120 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; el "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log
fi
if [ -z "$SKIP_PMC" ]; then
TAG=$TAG bash tools/gpu_pmc.sh regex1g > gpurun_out/${TAG}_pmc_regex1g.log 2>&1; el "pmc regex1g done"; tail -12 gpurun_out/${TAG}_pmc_regex1g.log | cut -c1-200
# the bench line below attaches profiles/${TAG}_*_pmc.json when the recorded source hash matches: put this run's there
cp gpurun_out/${TAG}_regex1g_pmc.json profiles/ 2>/dev/null
if [ -z "$SKIP_ENC_PMC" ]; then
SKIP_KT=1 TAG=$TAG bash tools/gpu_pmc.sh encode > gpurun_out/${TAG}_pmc_encode.log 2>&1; el "pmc encode done"; tail -4 gpurun_out/${TAG}_pmc_encode.log | cut -c1-200
cp gpurun_out/${TAG}_encode_pmc.json profiles/ 2>/dev/null
fi
for wl in $PMC_EXTRA; do   # secondary workloads' PMC passes (cfg2, basic1g): attached to their secondary entries
    SKIP_KT=1 TAG=$TAG bash tools/gpu_pmc.sh $wl > gpurun_out/${TAG}_pmc_$wl.log 2>&1; el "pmc $wl done"; tail -3 gpurun_out/${TAG}_pmc_$wl.log | cut -c1-200
    cp gpurun_out/${TAG}_${wl}_pmc.json profiles/ 2>/dev/null
done
fi
timeout -k 5 ${BENCH_TIMEOUT:-420} python bench.py ${BENCH_ARGS} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
el "bench rc=$?"; cut -c1-700 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
for x in $EXTRA; do
case $x in
dp1) BENCH_FORCE_DP=1 timeout -k 5 200 python bench.py --steps 2 --warmup 1 --secondary none --cpu-iters 0 > gpurun_out/${TAG}_bench_dp1_world1.json 2> gpurun_out/${TAG}_bench_dp1.err
     el "dp1 rc=$?"; cut -c1-400 gpurun_out/${TAG}_bench_dp1_world1.json ;;
big)  timeout -k 5 300 python bench.py --bytes 3900000000 --steps 1 --warmup 0 --secondary none --cpu-iters 0 > gpurun_out/${TAG}_big_3p9gb_bench.json 2> gpurun_out/${TAG}_big.err
      el "big rc=$?"; cut -c1-400 gpurun_out/${TAG}_big_3p9gb_bench.json; tail -2 gpurun_out/${TAG}_big.err ;;
iter) ITER_NPY=gpurun_out/${TAG}_regex1g_iter_us.npy timeout -k 5 200 python tools/iter_profile.py regex1g > gpurun_out/${TAG}_regex1g_iter_profile.json 2> gpurun_out/${TAG}_iter.err; el "iter rc=$?" ;;
esac
done
el "done"
