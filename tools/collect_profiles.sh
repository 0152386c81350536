#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench + PMC traffic of the dominant kernel -> gpurun_out/profiles_rNN/
set -e
R=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --warmup 1 --steps 1 --no_cpu_baseline > $OUT/bench_under_rocprof.json 2> /dev/null
# the warm-up pass carries the GEMM autotuner's timing launches: summarise the timed pass only
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt 60 --last-pass k_gather_codebook > $OUT/bench_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o pm -- python $GRAFT_REPO_ROOT/bench.py --n_timesteps 1 --warmup 0 --no_cpu_baseline --epochs 0 --epochs_exposure 1 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pm_FETCH_SIZE/pm_counter_collection.csv /tmp/pm_WRITE_SIZE/pm_counter_collection.csv k_flashILi40 > $OUT/flash40_traffic.json
cat $OUT/flash40_traffic.json; tail -1 $OUT/bench_under_rocprof.json | cut -c1-300
