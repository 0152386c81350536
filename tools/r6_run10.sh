#!/bin/bash
# round 6, GPU run 10: the final code -- whole -m gpu suite (durations), smoke() timed, default bench, the same under rocprofv3
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( time python -m pytest tests -m gpu -q -x --durations=25 -p no:cacheprovider ) > $O/gpu_tests_durations_final.log 2>&1
tail -4 $O/gpu_tests_durations_final.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -6 $O/smoke.log
bash tools/collect_final_r6.sh bench_only
