#!/bin/bash
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( python -m pytest tests/test_gpu_memflow.py tests/test_gpu_run.py -m gpu -q -x -s -p no:cacheprovider ) > $O/run12_tests.log 2>&1
grep -E "corr tiled|passed|failed|Error" $O/run12_tests.log | cut -c1-200
