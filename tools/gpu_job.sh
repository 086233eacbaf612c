#!/bin/bash
# One gpurun call = a list of steps, chosen by STEPS ("tests iter prof bench ..."), every step bounded
# by `timeout -k`, nothing reads stdin, outputs under gpurun_out/<TAG>_*.  Used as
#   gpurun --timeout 900 -- 'TAG=r3_a STEPS="tests iter" bash tools/gpu_job.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
TAG=${TAG:-job}
WL=${WL:-regex1g}
for step in $STEPS; do
case $step in
tests)
    timeout -k 5 ${TEST_TIMEOUT:-400} python -X faulthandler -m pytest ${PYTEST_FILES:-tests} -m gpu -q ${PYTEST_ARGS:--x} > gpurun_out/${TAG}_pytest.log 2>&1
    echo "pytest rc=$?"; tail -12 gpurun_out/${TAG}_pytest.log ;;
smoke)
    timeout -k 5 200 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/${TAG}_smoke.log ;;
iter)   # per-iteration device time, binned: ITER_OPTS="lean=0" etc.; several option sets separated by '|'
    IFS='|' read -ra SETS <<< "${ITER_OPTS:- }"
    n=0
    for o in "${SETS[@]}"; do
        timeout -k 5 300 python tools/iter_profile.py $WL $o > gpurun_out/${TAG}_iter_$n.json 2> gpurun_out/${TAG}_iter_$n.err
        echo "iter[$o] rc=$?"; tail -12 gpurun_out/${TAG}_iter_$n.err; n=$((n+1))
    done ;;
prof)   # rocprofv3 kernel trace of one train of the first PROF_MERGES merges
    rm -rf gpurun_out/${TAG}_prof
    timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o run -- python tools/train_n.py $WL ${PROF_MERGES:-31744} ${PROF_OPTS} > gpurun_out/${TAG}_prof.log 2>&1
    echo "prof rc=$?"; tail -2 gpurun_out/${TAG}_prof.log
    db=$(ls gpurun_out/${TAG}_prof/*/*.db gpurun_out/${TAG}_prof/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_stats.py $db > gpurun_out/${TAG}_kernel_stats.csv && head -20 gpurun_out/${TAG}_kernel_stats.csv
    [ -n "$db" ] && python tools/rocpd_phases.py $db > gpurun_out/${TAG}_phases.json
    [ -n "$db" ] && python tools/rocpd_timeline.py $db ${TIMELINE_N:-400} | head -${TIMELINE_HEAD:-120} > gpurun_out/${TAG}_timeline.txt
    rm -rf gpurun_out/${TAG}_prof ;;
pmc)    # kernel-trace stats + the two HBM-traffic counter passes, each its own run (tools/gpu_pmc.sh)
    TAG=$TAG bash tools/gpu_pmc.sh $WL ;;
bench)
    timeout -k 5 ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
    echo "bench rc=$?"; cut -c1-1500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err ;;
encode)
    timeout -k 5 600 python bench.py --workload encode --steps 3 --warmup 1 ${ENCODE_ARGS} > gpurun_out/${TAG}_encode.json 2> gpurun_out/${TAG}_encode.err
    echo "encode rc=$?"; cut -c1-1800 gpurun_out/${TAG}_encode.json; tail -3 gpurun_out/${TAG}_encode.err ;;
encprof)  # rocprofv3 kernel trace of the encode workload
    rm -rf gpurun_out/${TAG}_eprof
    timeout -k 5 500 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_eprof -o run -- python bench.py --workload encode --steps 2 --warmup 1 --cpu-iters 0 > gpurun_out/${TAG}_eprof.log 2>&1
    echo "encprof rc=$?"
    db=$(ls gpurun_out/${TAG}_eprof/*/*.db gpurun_out/${TAG}_eprof/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_stats.py $db > gpurun_out/${TAG}_encode_kernel_stats.csv && grep -i "k_enc\|k_scan\|k_encode" gpurun_out/${TAG}_encode_kernel_stats.csv | sed 's/(.*)"/"/' | cut -c1-120
    rm -rf gpurun_out/${TAG}_eprof ;;
py)     # PY="tools/x.py args": any bounded python step
    timeout -k 5 ${PY_TIMEOUT:-600} python $PY > gpurun_out/${TAG}_py.log 2>&1; echo "py rc=$?"; tail -${PY_TAIL:-30} gpurun_out/${TAG}_py.log ;;
esac
done
