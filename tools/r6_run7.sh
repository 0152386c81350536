#!/bin/bash
# round 6, GPU run 7: depthwise convolutions on v_fma_mix_f32 -- tests, frame-pair time, kernel table
set -x
O=gpurun_out/profiles_r6; mkdir -p $O
( time python -m pytest tests/test_gpu_memflow.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider ) > $O/run7_tests.log 2>&1
tail -3 $O/run7_tests.log
for i in 1 2; do timeout 600 python tools/micro/prof_producers.py --what memflow 2>/dev/null | tail -1; done > $O/memflow_after_dwconv.txt
cut -c1-200 $O/memflow_after_dwconv.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ktp -o kt -- python $GRAFT_REPO_ROOT/tools/micro/prof_producers.py --what memflow > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ktp 24 > $GRAFT_REPO_ROOT/$O/memflow_kernel_stats.txt
head -8 $GRAFT_REPO_ROOT/$O/memflow_kernel_stats.txt
