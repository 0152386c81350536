#!/bin/bash
# round 5, second pass: grid of the flag-gated exact pass with the windowed flag scan, IN THE PASS (60 frames, matching chain on its side stream), same box
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -q -m gpu -k "attention" 2>&1 | tail -2
for g in 512 32 512 32 8; do
  TCL_FLASH_GATE_BLOCKS=$g timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('gate=$g', round(r['value'],4), r['phase_seconds'])"
done
