#!/bin/bash
# GPU box (via gpurun): final evidence of round 5 (after the path-2 half of the second pass) -> gpurun_out/profiles_r5c/
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r5c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_path2_dist.py -m gpu -q -k "config5 or path2 or stage" > $OUT/gpu_tests_path2_fullsize.log 2>&1; tail -3 $OUT/gpu_tests_path2_fullsize.log
timeout 1200 python bench.py > $OUT/bench_final_run.json 2> $OUT/bench_final_run.err; tail -c 300 $OUT/bench_final_run.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --profile_steps 0 > $OUT/bench_under_rocprof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt 70 --timed-pass > $OUT/bench_kernel_stats.txt; rm -rf /tmp/kt
head -12 $OUT/bench_kernel_stats.txt | cut -c1-150
