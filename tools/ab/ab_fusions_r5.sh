#!/bin/bash
# round 5: the two element-wise fusions of the round (GroupNorm apply writes the raw concat: TCL_GN_CONCAT; attn2 to_q straight into the query panel:
# TCL_QPANEL) on / off, 60-frame pass of the metric's clip, two interleaved runs each, same box
for i in 1 2; do
  for t in 0 1; do
    TCL_GN_CONCAT=$t TCL_QPANEL=$t python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('fusions=$t', round(r['value'],4), r['phase_seconds'])"
  done
done
