#!/bin/bash
# round 6, GPU run 8: split-KV memory-read attention -- tests, A/B over the number of chunks, kernel table
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( time python -m pytest tests/test_gpu_memflow.py tests/test_gpu_run.py -m gpu -q -x -s -p no:cacheprovider ) > $O/run8_tests.log 2>&1
grep -E "split-KV|passed|failed|Error|memflow engine" $O/run8_tests.log | cut -c1-200
for i in 1 2; do for n in 0 2 3 5 6; do
  echo "== TCL_MEMFLOW_SPLITKV=$n"
  TCL_MEMFLOW_SPLITKV=$n timeout 600 python tools/micro/prof_producers.py --what memflow 2>/dev/null | tail -1 | cut -c1-140
done; done > $O/ab_splitkv.txt 2>&1
grep -v "^+" $O/ab_splitkv.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ktp -o kt -- python $GRAFT_REPO_ROOT/tools/micro/prof_producers.py --what memflow > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ktp 30 > $GRAFT_REPO_ROOT/$O/memflow_kernel_stats.txt
head -10 $GRAFT_REPO_ROOT/$O/memflow_kernel_stats.txt
