#!/bin/bash
# round 5, second pass, whole effect: library of commit e04276a (the first session's final code) vs the tree's, 60-frame pass of the metric's clip, two interleaved runs each, same box
BASE=$PWD/tc_light_amd/libtclight_hip_base.so; NEW=$PWD/tc_light_amd/libtclight_hip.so
for i in 1 2; do for l in base new; do
  p=$BASE; [ $l = new ] && p=$NEW
  TCL_LIB_PATH=$p timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$l', round(r['value'],4), r['phase_seconds'])"
done; done
