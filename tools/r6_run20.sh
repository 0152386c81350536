#!/bin/bash
# round 6, GPU run 20: the skewed two-group schedule (TCL_SKEW=1) beside the round-6 chain, 60 frames, two interleaved runs each
O=gpurun_out/profiles_r6; mkdir -p $O
for i in 1 2; do for t in 0 1; do
  TCL_SKEW=$t timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 1 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('TCL_SKEW=$t', round(r['value'],4), r['phase_seconds'], 'flash', round(r['roofline']['achieved']), 'match', round(r['roofline_match']['achieved']), 'gemm', round(r['roofline_gemm']['achieved']))"
done; done > $O/ab_skew_r6.txt 2>&1
grep "^TCL" $O/ab_skew_r6.txt
