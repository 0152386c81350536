#!/bin/bash
# usage: pmc_run2.sh <script.py> <kernel substring> ; TCC counter passes
cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmct_$i -o p -- python $GRAFT_REPO_ROOT/$1 > /tmp/pmct_$i.log 2>&1 || tail -5 /tmp/pmct_$i.log
done
python $GRAFT_REPO_ROOT/tools/micro/pmc_agg.py "$2" /tmp/pmct_*/p_counter_collection.csv
