#!/bin/bash
# round 6, GPU run 3: the box's CPU allowance, the carried-metric matching chain (bit-identity tests + 60-frame A/B), the e2e oracle legs under a 12-thread budget
set -x
O=gpurun_out
( nproc; python -c "import os;print('cpu_count',os.cpu_count(),'affinity',len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; lscpu | head -20 ) > $O/r6_cpuinfo.txt 2>&1
( time python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py tests/test_gpu_skew.py tests/test_gpu_config4.py -m gpu -q -x -p no:cacheprovider --durations=10 ) > $O/r6_tome_tests.log 2>&1
tail -3 $O/r6_tome_tests.log
( time python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "tome or unet_pass or config3" ) > $O/r6_tome_tests_full.log 2>&1
tail -3 $O/r6_tome_tests_full.log
for i in 1 2; do for t in 1 0; do
  TCL_TOME_CAT=$t timeout 900 python bench.py --frames 60 --no_cpu_baseline --no_extras --profile_steps 0 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('TCL_TOME_CAT=$t', round(r['value'],4), r['phase_seconds'])"
done; done > $O/r6_ab_tome_carried.txt 2>&1
cat $O/r6_ab_tome_carried.txt
( time TCL_E2E_WAIT=900 TCL_TEST_CEILING=900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --durations=0 -p no:cacheprovider ) > $O/r6_e2e_b.log 2>&1
grep -E "oracle leg|passed|failed" $O/r6_e2e_b.log | cut -c1-200
