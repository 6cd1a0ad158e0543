#!/bin/bash
# PMC passes (one counter set per run) for one conv primitive: bash tools/pmc.sh <tag> <one_conv args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_$1; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $O -o p$i -- python3 $R/tools/one_conv.py "$@" > $O/log$i.txt 2>&1 || { echo "pass $i failed"; tail -3 $O/log$i.txt; exit 1; }
done
ls $O
