python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -x -q -k "unet or forward_many" 2>&1 | tail -2
for v in 1 2 1 2; do echo == ATTN STREAMS $v; TCL_ATTN_STREAMS=$v python bench.py --no_extras --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_seconds'])"; done
