#!/bin/bash
# GPU box (via gpurun): final evidence of round 5's second pass -> gpurun_out/profiles_r5b/
#   1. the plain default bench (the metric's configuration, CPU sample included)   2. kernel-trace stats of its timed pass   3. the GPU test suite
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r5b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > $OUT/bench_final_run.json 2> $OUT/bench_final_run.err; tail -c 600 $OUT/bench_final_run.json | head -c 600; echo
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --profile_steps 0 > $OUT/bench_under_rocprof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt 70 --timed-pass > $OUT/bench_kernel_stats.txt; rm -rf /tmp/kt
head -25 $OUT/bench_kernel_stats.txt | cut -c1-150
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
