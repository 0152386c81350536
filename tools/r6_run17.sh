#!/bin/bash
# round 6, GPU run 17: single-block radix select (k_thr_select1) -- map tests, then 60-frame A/B TCL_THR1=0/1, two interleaved runs each
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -q -x -p no:cacheprovider -k "tome or vidtome or unet" ) > $O/run17_tests.log 2>&1; tail -n 2 $O/run17_tests.log
for i in 1 2; do for t in 0 1; do
  TCL_THR1=$t timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 1 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('TCL_THR1=$t', round(r['value'],4), r['phase_seconds'], 'flash', round(r['roofline']['achieved']), 'match', round(r['roofline_match']['achieved']), 'gemm', round(r['roofline_gemm']['achieved']))"
done; done > $O/ab_thr1.txt 2>&1
grep "^TCL" $O/ab_thr1.txt
