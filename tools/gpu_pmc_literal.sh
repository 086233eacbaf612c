#!/bin/bash
# the literal path (mode=0) on the headline input: kernel trace, then one PMC pass of the same command (each its own run)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
M=${1:-192}
CMD="python tools/train_n.py regex1g $M mode=0"
rm -rf gpurun_out/lit_kt gpurun_out/lit_r
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d gpurun_out/lit_kt -o run -- $CMD > gpurun_out/lit_kt.log 2>&1; echo "kt rc=$?"
DB=$(find gpurun_out/lit_kt -name "*.db" | head -1)
python tools/rocpd_stats.py "$DB" > gpurun_out/r6_literal_kernel_stats.csv; rm -rf gpurun_out/lit_kt
timeout -k 5 600 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_ATOMIC_32B --kernel-trace -d gpurun_out/lit_r -o run -- $CMD > gpurun_out/lit_r.log 2>&1; echo "pmc rc=$?"
R=$(find gpurun_out/lit_r -name "*.db" | head -1)
python tools/pmc_literal.py "$R" gpurun_out/r6_literal_kernel_stats.csv $M gpurun_out/r6_literal_pmc.json; echo "summary rc=$?"
rm -rf gpurun_out/lit_r
