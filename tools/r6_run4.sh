#!/bin/bash
# round 6, GPU run 4: the whole -m gpu suite in its new order (timing + durations), stage-2 run-ahead A/B, producers profile, K=8 MFMA issue cost
set -x
O=gpurun_out; mkdir -p $O/profiles_r6
( time python -m pytest tests -m gpu -q -x --durations=25 -p no:cacheprovider ) > $O/profiles_r6/gpu_tests_durations.log 2>&1
tail -4 $O/profiles_r6/gpu_tests_durations.log
for i in 1 2; do for t in 0 1; do
  echo "== TCL_ADAM_RUNAHEAD=$t"
  P2_STAGES=2 P2_DIGEST=1 TCL_ADAM_RUNAHEAD=$t timeout 600 python tools/micro/bench_p2.py 300 720 1280 95 0.02 2>&1 | grep "^stage"
done; done > $O/profiles_r6/ab_path2_runahead.txt 2>&1
cat $O/profiles_r6/ab_path2_runahead.txt
timeout 900 python tools/micro/prof_producers.py > $O/profiles_r6/producers.json 2> $O/profiles_r6/producers.err
cat $O/profiles_r6/producers.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ktp -o kt -- python $GRAFT_REPO_ROOT/tools/micro/prof_producers.py --what memflow > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ktp 40 > $GRAFT_REPO_ROOT/$O/profiles_r6/memflow_kernel_stats.txt
rm -rf /tmp/ktp
rocprofv3 --kernel-trace --stats -d /tmp/ktr -o kt -- python $GRAFT_REPO_ROOT/tools/micro/prof_producers.py --what rmbg > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ktr 30 > $GRAFT_REPO_ROOT/$O/profiles_r6/rmbg_kernel_stats.txt
rm -rf /tmp/ktr
cd $GRAFT_REPO_ROOT
head -30 $O/profiles_r6/memflow_kernel_stats.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && /tmp/valu_rates > $O/profiles_r6/valu_rates.txt 2>&1
tail -8 $O/profiles_r6/valu_rates.txt
