#!/bin/bash
# round 6, GPU run 11: banded tile correlation (tests + frame-pair time), then the C = 640 strip kernel beside the round-6 chain (TCL_TOME640), 60 frames, two interleaved runs each
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( python -m pytest tests/test_gpu_memflow.py -m gpu -q -x -s -p no:cacheprovider ) > $O/run11_tests.log 2>&1
grep -E "corr tiled|passed|failed|Error" $O/run11_tests.log | cut -c1-200
for i in 1 2; do timeout 600 python tools/micro/prof_producers.py --what memflow 2>/dev/null | tail -1 | cut -c1-120; done > $O/memflow_after_bands.txt; cat $O/memflow_after_bands.txt
for i in 1 2; do for t in 0 1; do
  TCL_TOME640=$t timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('TCL_TOME640=$t', round(r['value'],4), r['phase_seconds'])"
done; done > $O/ab_tome640_r6.txt 2>&1
grep "^TCL" $O/ab_tome640_r6.txt
