for v in 0 7 0 7; do echo == TPB $v; TCL_FLASH_TPB=$v python tools/micro/bench_attn.py 2>&1 | grep "d="; done
TCL_FLASH_TPB=7 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attn or flash or attention" 2>&1 | tail -2
