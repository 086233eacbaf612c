#!/bin/bash
# kernel-trace stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only)
# of one bench step of workload $1 (default regex1g); summaries land in gpurun_out/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
WL=${1:-regex1g}
CMD="python bench.py --workload $WL --steps 1 --warmup 0 --secondary none --cpu-iters 0"
rm -rf gpurun_out/prof_kt gpurun_out/prof_f gpurun_out/prof_w
timeout -k 5 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o run -- $CMD > gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > gpurun_out/${TAG:-r3}_${WL}_kernel_stats.csv && rm -rf gpurun_out/prof_kt
timeout -k 5 700 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_f -o run -- $CMD > gpurun_out/prof_f.log 2>&1; echo "fetch rc=$?"
timeout -k 5 700 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_w -o run -- $CMD > gpurun_out/prof_w.log 2>&1; echo "write rc=$?"
F=$(find gpurun_out/prof_f -name "*.db" | head -1); W=$(find gpurun_out/prof_w -name "*.db" | head -1)
NB=1000000000; [ "$WL" = "cfg2" ] && NB=100000000
python tools/pmc_summary.py "$F" "$W" $WL gpurun_out/${TAG:-r3}_${WL}_pmc.json $NB; echo "summary rc=$?"
rm -rf gpurun_out/prof_f gpurun_out/prof_w
head -8 gpurun_out/${TAG:-r3}_${WL}_kernel_stats.csv | cut -c1-150
tail -1 gpurun_out/prof_f.log | cut -c1-300
