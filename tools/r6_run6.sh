#!/bin/bash
# round 6, GPU run 6: the tile-sharing correlation lookup -- tests, then a same-box A/B of a MemFlowNet frame pair, then its kernel table
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( time python -m pytest tests/test_gpu_memflow.py tests/test_gpu_run.py -m gpu -q -x -s -p no:cacheprovider ) > $O/run6_tests.log 2>&1
grep -E "corr tiled|passed|failed|Error" $O/run6_tests.log | cut -c1-200
for i in 1 2; do for t in 0 1; do
  echo "== TCL_CORR_TILED=$t"
  TCL_CORR_TILED=$t timeout 600 python tools/micro/prof_producers.py --what memflow 2>/dev/null | tail -1
done; done > $O/ab_corr_tiled.txt 2>&1
grep -v "^+" $O/ab_corr_tiled.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ktp -o kt -- python $GRAFT_REPO_ROOT/tools/micro/prof_producers.py --what memflow > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ktp 24 > $GRAFT_REPO_ROOT/$O/memflow_kernel_stats.txt
head -12 $GRAFT_REPO_ROOT/$O/memflow_kernel_stats.txt
