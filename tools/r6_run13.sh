#!/bin/bash
# round 6, GPU run 13: the configs[0] end-to-end tests with ALL oracle legs (TCL_E2E_FULL=1: the whole-path f16 floor and the whole-path stage 1/2 as well)
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( time TCL_E2E_FULL=1 TCL_E2E_WAIT=1200 TCL_TEST_CEILING=1300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --durations=0 -p no:cacheprovider ) > $O/e2e_full.log 2>&1
grep -E "e2e config|oracle leg|passed|failed|s call" $O/e2e_full.log | cut -c1-400
