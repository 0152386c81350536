#!/bin/bash
# round 5 flash lab: product (cur) vs DPV=48 panel vs static priority for alternate blocks; attention tests under the dpv48 build
for r in 1 2; do for v in cur dpv48 prio1; do
  if [ "$v" = cur ]; then unset TCL_LIB_PATH; else export TCL_LIB_PATH=$PWD/scratch/libtclight_$v.so; fi
  echo "== $v (round $r)"; python tools/micro/bench_attn_long.py 2>&1 | grep "n="; python tools/micro/bench_attn.py 2>&1 | grep "d=40"
done; done
export TCL_LIB_PATH=$PWD/scratch/libtclight_dpv48.so
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -3
