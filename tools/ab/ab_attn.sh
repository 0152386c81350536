#!/bin/bash
# A/B of attention builds: scratch/libtclight_<tag>.so (whole library with another attn.o) against the in-tree one; two interleaved rounds
for r in 1 2; do
  for v in "$@" cur; do
    if [ "$v" = cur ]; then unset TCL_LIB_PATH; else export TCL_LIB_PATH=$PWD/scratch/libtclight_$v.so; fi
    echo "== $v (round $r)"; python tools/micro/bench_attn.py 2>&1 | grep "d="
  done
done
