#!/bin/bash
# round 5: attn1's QKV projection -> attention panels in the GEMM epilogue (TCL_QKV_PANEL=1, default) vs GEMM + k_pack_qkv (0); 60-frame pass, same box
for i in 1 2; do
  for t in 0 1; do
    TCL_QKV_PANEL=$t python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('qkv_panel=$t', round(r['value'],4), r['phase_seconds'], r['max_memory_allocated_MiB'])"
  done
done
