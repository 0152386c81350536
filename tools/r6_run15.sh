#!/bin/bash
# round 6, GPU run 15: the per-chunk QKV projection on the side stream (TCL_QKV_SIDE) beside the round-6 chain, 60 frames, two interleaved runs each
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
for i in 1 2; do for t in 0 1; do
  TCL_QKV_SIDE=$t timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 1 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('TCL_QKV_SIDE=$t', round(r['value'],4), r['phase_seconds'], 'flash', round(r['roofline']['achieved']), 'match', round(r['roofline_match']['achieved']), 'gemm', round(r['roofline_gemm']['achieved']))"
done; done > $O/ab_qkv_side_r6.txt 2>&1
grep "^TCL" $O/ab_qkv_side_r6.txt
( python -m pytest tests/test_gpu_memflow.py -m gpu -q -x -p no:cacheprovider ) 2>&1 | tail -2
