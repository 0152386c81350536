#!/bin/bash
# Regenerate tc_light_amd/gemm_tune_gfx950.txt on an MI355X: every GEMM / conv shape of the metric's pass (300 x 1280 x 720) and of
# BASELINE configs[1] (30 x 960 x 720) and configs[3] (60 x 960 x 720, VidToMe 0.9 / 0.8) is timed on first use over the valid tiles (csrc/gemm.hip tile_ok) and the winners are saved.
# Usage (GPU box): tools/retune_gemm_table.sh   -> gpurun_out/gemm_tune_gfx950.txt (copy it over the committed table)
set -e
export TCL_GEMM_NEAR=0        # measure every shape: no tile borrowed from a nearest-M twin (csrc/gemm.hip)
mkdir -p gpurun_out
OUT=gpurun_out/gemm_tune_gfx950.txt
rm -f $OUT
TCL_GEMM_TABLE=/nonexistent python bench.py --steps 20 --warmup 0 --profile_steps 0 --no_cpu_baseline --no_extras --save_gemm_table $OUT > gpurun_out/retune_pass300.json 2> gpurun_out/retune_pass300.err
TCL_GEMM_TABLE=$OUT python bench.py --frames 30 --height 720 --width 960 --steps 20 --warmup 0 --profile_steps 0 --no_cpu_baseline --no_extras --save_gemm_table $OUT > gpurun_out/retune_pass30.json 2> gpurun_out/retune_pass30.err
TCL_GEMM_TABLE=$OUT python tools/retune_config3.py $OUT > gpurun_out/retune_config3.txt 2> gpurun_out/retune_config3.err
wc -l $OUT
