#!/bin/bash
# round 5, second pass, path 2: loads pinned ahead of their guards (k_flow_loss tap gathers, k_pixel_losses, k_pool2, k_ssim_* staging, k_gather_codebook,
# k_codebook_bwd).  base = library of commit e04276a.  Stage 1 / stage 2 alone, 300 x 1280 x 720, outputs digested (must agree), then per-kernel times.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5b; mkdir -p $OUT
BASE=$PWD/tc_light_amd/libtclight_hip_base.so; NEW=$PWD/tc_light_amd/libtclight_hip.so
for reuse in 0.02 0.7; do for l in base new base new; do
  p=$BASE; [ $l = new ] && p=$NEW
  echo "== $l reuse=$reuse"
  P2_DIGEST=1 TCL_LIB_PATH=$p timeout 600 python tools/micro/bench_p2.py 300 720 1280 48 $reuse 2>&1 | grep "^stage"
done; done
bash tools/ab/prof_p2_libs.sh 0.02 _base ""
timeout 900 python -m pytest tests/test_gpu_path2.py tests/test_gpu_path2_dist.py -x -q -m gpu 2>&1 | tail -3
