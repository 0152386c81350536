python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -x -q -k "unet or forward_many" 2>&1 | tail -2
for v in 0 1 0 1; do echo == ATTN HI $v; TCL_ATTN_HI=$v python bench.py --no_extras --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['phase_seconds']['denoise'])"; done
