#!/bin/bash
# GPU box (via gpurun): the round-6 evidence set -> gpurun_out/profiles_r6/
#   1. the default bench (the metric's configuration) -> bench_final_run.json
#   2. the same command under rocprofv3 --kernel-trace --stats, timed pass only -> bench_kernel_stats.txt
#   3. [traffic] HBM-side bytes per head_dim-40 attention call in the pass (separate --pmc passes) and per stage-2 iteration (tools/collect_path2_traffic.sh)
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r6
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$1" != traffic_only ]; then
timeout 1500 python bench.py > $OUT/bench_final_run.json 2> $OUT/bench_final_run.err; tail -c 400 $OUT/bench_final_run.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --profile_steps 0 > $OUT/bench_under_rocprof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt 70 --timed-pass > $OUT/bench_kernel_stats.txt; rm -rf /tmp/kt
head -14 $OUT/bench_kernel_stats.txt | cut -c1-150
fi
if [ "$1" != bench_only ]; then
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  for try in 1 2; do
    TCL_TOME_STREAM=0 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o pm -- python $GRAFT_REPO_ROOT/bench.py --frames 60 --steps 2 --warmup 0 --no_cpu_baseline --no_extras --epochs 0 --epochs_exposure 1 --profile_steps 0 > /tmp/pm_$c.log 2>&1 && break
    echo "pass $c try $try failed"; tail -3 /tmp/pm_$c.log
  done
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pm_FETCH_SIZE/pm_counter_collection.csv /tmp/pm_WRITE_SIZE/pm_counter_collection.csv k_flashILi40 k_flashILi40ELi48ELi64ELi2ELi4ELi2ELi0ELi0E > $OUT/flash40_traffic.json
cat $OUT/flash40_traffic.json
bash $GRAFT_REPO_ROOT/tools/collect_path2_traffic.sh r6 38
fi
