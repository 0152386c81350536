set -x
export TCL_E2E_WAIT=400 TCL_TEST_CEILING=500
( time python -m pytest tests/test_gpu_rccl.py -m gpu -q -s -p no:cacheprovider ) > gpurun_out/r6_rccl.log 2>&1
for t in 8 16 32 64; do
  ( time TCL_TEST_THREADS=$t python -m pytest tests/test_gpu_denoise_loop.py tests/test_gpu_e2e.py::test_multi_axis_bank_carry_over "tests/test_gpu_unet.py" -m gpu -q --durations=8 -p no:cacheprovider ) > gpurun_out/r6_threads_$t.log 2>&1
done
( time python -m pytest tests/test_gpu_e2e.py tests/test_gpu_e2e_dist.py -m gpu -q -s --durations=0 -p no:cacheprovider ) > gpurun_out/r6_e2e.log 2>&1
tail -5 gpurun_out/r6_e2e.log
