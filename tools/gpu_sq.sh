#!/bin/bash
# SQ counter passes of ONE train (tools/train_n.py regex1g 31744 $OPTS), reduced per kernel and per phase of training
# (tools/pmc_sq_phases.py).  Each pass is its own rocprofv3 run with --kernel-trace only.  TAG = output prefix.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
exec </dev/null
TAG=${TAG:-r6_sq}
KERN=${KERN:-k_merge_chain}
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > gpurun_out/${TAG}_counters_available.txt
wc -l gpurun_out/${TAG}_counters_available.txt
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT"; do
  i=$((i+1))
  rm -rf gpurun_out/sq_$i
  timeout -k 5 400 rocprofv3 --pmc $SET --kernel-trace -d gpurun_out/sq_$i -o run -- python tools/train_n.py regex1g 31744 $OPTS > gpurun_out/sq_$i.log 2>&1; echo "sq$i rc=$?"
  db=$(find gpurun_out/sq_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/pmc_sq_phases.py $db $KERN > gpurun_out/${TAG}_pass$i.json; echo "reduce rc=$?"; else tail -5 gpurun_out/sq_$i.log | cut -c1-300; fi
  rm -rf gpurun_out/sq_$i
done
head -c 3000 gpurun_out/${TAG}_pass1.json
