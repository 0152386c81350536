#!/bin/bash
# round 5: stage 2's lazy Adam with ONE visit per row and iteration (TCL_ADAM_LAZY_V1=0, opt-in) vs rounds 3-4's catch-up launch + gather + step (default),
# bench codebook regime (reuse 0.02: K ~ N H W) and realistic track lengths (0.7: dense Adam, unaffected), 300 x 1280 x 720, 60 iterations
for r in 1 2; do for v in 0 1; do
  echo "== TCL_ADAM_LAZY_V1=$v (round $r)"
  TCL_ADAM_LAZY_V1=$v python tools/micro/bench_p2.py 300 720 1280 60 0.02 2>&1 | grep "^stage"
done; done
echo "== realistic codebook (dense Adam)"; python tools/micro/bench_p2.py 300 720 1280 60 0.7 2>&1 | grep "^stage"
