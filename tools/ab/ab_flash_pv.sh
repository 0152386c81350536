#!/bin/bash
# round 5: head_dim-40 PV on 16x16x32 (+ permlane swaps) vs 32x32x16; two interleaved rounds + the attention tests under PV32
mkdir -p gpurun_out/r5
for r in 1 2; do for pv in 16 32; do echo "== TCL_FLASH_PV=$pv (round $r)"; TCL_FLASH_PV=$pv python tools/micro/bench_attn.py 2>&1 | grep "d=40"; done; done
echo "== long"
for pv in 16 32; do echo "-- pv $pv"; TCL_FLASH_PV=$pv python tools/micro/bench_attn_long.py 2>&1 | tail -4; done
