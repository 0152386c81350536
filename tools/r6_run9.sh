#!/bin/bash
# round 6, GPU run 9: k_tome_match320 with two src sub-tiles per wave (TCL_TOME320=2) -- bit tests, alone, in the pass
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( TCL_TOME320=2 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "tome or vidtome" ) > $O/run9_tests_x2.log 2>&1; tail -2 $O/run9_tests_x2.log
( python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "tome or vidtome" ) > $O/run9_tests_x1.log 2>&1; tail -2 $O/run9_tests_x1.log
for t in 4 2 4 2; do echo "== TCL_TOME320=$t"; TCL_TOME320=$t python tools/micro/bench_tome.py 2>/dev/null | grep "C=320"; done > $O/ab_tome_x2_alone.txt 2>&1
grep -v "^+" $O/ab_tome_x2_alone.txt
for i in 1 2; do for t in 4 2; do
  TCL_TOME320=$t timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 1 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('TCL_TOME320=$t', round(r['value'],4), r['phase_seconds'], 'flash', round(r['roofline']['achieved']), 'match', round(r['roofline_match']['achieved']), 'gemm', round(r['roofline_gemm']['achieved']))"
done; done > $O/ab_tome_x2_inpass.txt 2>&1
grep -v "^+" $O/ab_tome_x2_inpass.txt
