#!/bin/bash
# round 5: grid of the flag-gated exact pass behind the speculative flash kernel (blocks that walk the flags), sustained call rate at T = 35 640
for r in 1 2; do for g in 512 64 16; do echo "== TCL_FLASH_GATE_BLOCKS=$g (round $r)"; TCL_FLASH_GATE_BLOCKS=$g python tools/micro/bench_attn_long.py 2>&1 | grep "n="; done; done
TCL_FLASH_GATE_BLOCKS=16 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "attention" 2>&1 | tail -2
