#!/bin/bash
# round 5, second pass: memory-level parallelism in the HBM-bound kernels (several loads in flight per thread: k_gn_stats / k_gn_apply / k_layernorm /
# k_tome_normalize; the stage-2 gather kernels k_codebook_bwd / k_adam_catchup_frame / k_adam_touched_frame walk their dependent chain once per 4 pixels).
# base = the library built from the commit before (tc_light_amd/libtclight_hip_base.so), new = the tree's.  Same box throughout.
OUT=gpurun_out/r5b; mkdir -p $OUT
BASE=$PWD/tc_light_amd/libtclight_hip_base.so; NEW=$PWD/tc_light_amd/libtclight_hip.so
echo "### 1. bit comparison + timing of the element-wise kernels" | tee $OUT/cmp_elem.txt
timeout 600 python tools/ab/cmp_elem_libs.py $BASE $NEW 2>&1 | tee -a $OUT/cmp_elem.txt
echo "### 2. stage 1 / stage 2 alone, outputs digested" | tee $OUT/path2.txt
for reuse in 0.02 0.7; do for l in base new base new; do
  p=$BASE; [ $l = new ] && p=$NEW
  echo "== $l reuse=$reuse" | tee -a $OUT/path2.txt
  P2_DIGEST=1 TCL_LIB_PATH=$p timeout 600 python tools/micro/bench_p2.py 300 720 1280 48 $reuse 2>&1 | grep "^stage" | tee -a $OUT/path2.txt
done; done
echo "### 3. tests of the touched kernels" | tee $OUT/tests.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path2.py tests/test_gpu_unet.py tests/test_gpu_vae.py -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT/tests.txt
echo "### 4. 60-frame pass A/B" | tee $OUT/ab60.txt
for i in 1 2; do for l in base new; do
  p=$BASE; [ $l = new ] && p=$NEW
  TCL_LIB_PATH=$p timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$l', round(r['value'],4), r['phase_seconds'])" | tee -a $OUT/ab60.txt
done; done
