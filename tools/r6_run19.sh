#!/bin/bash
# round 6, GPU run 19: the whole -m gpu suite and smoke() on the final commit
O=gpurun_out/profiles_r6; mkdir -p $O
( time python -m pytest tests -m gpu -q -x --durations=25 -p no:cacheprovider ) > $O/gpu_tests_durations_final.log 2>&1
tail -n 4 $O/gpu_tests_durations_final.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -n 5 $O/smoke.log
