#!/bin/bash
# kernel-trace stats + ONE PMC pass (the three raw, size-weighted TCC/EA counters: tools/pmc_summary.py) of one bench
# step of workload $1 (default regex1g), each its own rocprofv3 run (kernel trace only); summaries land in gpurun_out/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
WL=${1:-regex1g}
CMD="python bench.py --workload $WL --steps 1 --warmup 0 --secondary none --cpu-iters 0"
[ "$WL" = "encode" ] && CMD="python bench.py --workload encode --steps 1 --warmup 0 --cpu-iters 0"
rm -rf gpurun_out/prof_kt gpurun_out/prof_r
if [ -z "$SKIP_KT" ]; then
timeout -k 5 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o run -- $CMD > gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > gpurun_out/${TAG:-r4}_${WL}_kernel_stats.csv && rm -rf gpurun_out/prof_kt
fi
timeout -k 5 700 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_ATOMIC_32B --kernel-trace -d gpurun_out/prof_r -o run -- $CMD > gpurun_out/prof_r.log 2>&1; echo "pmc rc=$?"
R=$(find gpurun_out/prof_r -name "*.db" | head -1)
NB=1000000000; NM=31744; [ "$WL" = "cfg2" ] && NB=100000000 && NM=3840
if [ "$WL" = "encode" ]; then
python tools/pmc_encode_summary.py "$R" gpurun_out/${TAG:-r4}_encode_pmc.json; echo "summary rc=$?"
else
PEAK=gpurun_out/${TAG:-r4}_atomic_peak.json; [ -f "$PEAK" ] || PEAK=profiles/r6_atomic_peak.json
python tools/pmc_summary.py "$R" $WL gpurun_out/${TAG:-r4}_${WL}_pmc.json $NB $NM gpurun_out/${TAG:-r4}_${WL}_kernel_stats.csv $PEAK; echo "summary rc=$?"
fi
rm -rf gpurun_out/prof_r
head -8 gpurun_out/${TAG:-r4}_${WL}_kernel_stats.csv 2>/dev/null | cut -c1-150
tail -1 gpurun_out/prof_r.log | cut -c1-300
