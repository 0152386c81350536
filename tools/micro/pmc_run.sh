#!/bin/bash
# usage: pmc_run.sh <script.py> <kernel substring> ; SQ counter passes
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcq_$i -o p -- python $GRAFT_REPO_ROOT/$1 > /tmp/pmcq_$i.log 2>&1 || tail -5 /tmp/pmcq_$i.log
done
python $GRAFT_REPO_ROOT/tools/micro/pmc_agg.py "$2" /tmp/pmcq_*/p_counter_collection.csv
