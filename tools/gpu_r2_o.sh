#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
rm -rf gpurun_out/prof_sq
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d gpurun_out/prof_sq -o run -- python tools/train_n.py regex1g 120 > gpurun_out/prof_sq.log 2>&1; echo "sq rc=$?"
DB=$(find gpurun_out/prof_sq -name "*.db" | head -1); python tools/pmc_sq.py "$DB" k_merge_ab_dense; rm -rf gpurun_out/prof_sq
rm -rf gpurun_out/prof_sq2
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR --kernel-trace -d gpurun_out/prof_sq2 -o run -- python tools/train_n.py regex1g 120 > gpurun_out/prof_sq2.log 2>&1; echo "sq2 rc=$?"
DB=$(find gpurun_out/prof_sq2 -name "*.db" | head -1); python tools/pmc_sq.py "$DB" k_merge_ab_dense; rm -rf gpurun_out/prof_sq2
tail -3 gpurun_out/prof_sq2.log | cut -c1-300
