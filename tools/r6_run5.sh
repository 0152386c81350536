#!/bin/bash
# round 6, GPU run 5: MemFlowNet step graphs + 2-wave memory-read attention (tests, then A/B), persistent-panel guard
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( time python -m pytest tests/test_gpu_memflow.py tests/test_gpu_unet.py tests/test_gpu_run.py tests/test_gpu_rccl.py -m gpu -q -x -p no:cacheprovider --durations=8 ) > $O/run5_tests.log 2>&1
tail -12 $O/run5_tests.log
for i in 1 2; do
for cfg in "TCL_FLASH128_NW=4 TCL_MEMFLOW_GRAPH=0" "TCL_FLASH128_NW=2 TCL_MEMFLOW_GRAPH=0" "TCL_FLASH128_NW=2 TCL_MEMFLOW_GRAPH=1"; do
  echo "== $cfg"
  env $cfg timeout 600 python tools/micro/prof_producers.py --what memflow 2>/dev/null | tail -1
done; done > $O/ab_memflow.txt 2>&1
cat $O/ab_memflow.txt
