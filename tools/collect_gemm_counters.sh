#!/bin/bash
# GPU box: SQ counters + HBM-side traffic of the GEMM family -> gpurun_out/profiles_r3/  (evidence for k_gemm8* and k_gemm_dma)
#   8-wave kernel: the stand-alone lab on conv 640->320 @90x160 x64 (shape 22) and lin 368640x640x2560 (shape 19), schedule 2 = k_gemm8p
#   4-wave kernel: tools/micro/pmc_gemm_dma.py (short-K Linears)
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/tools/micro/bin/gemm8_lab
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16" "FETCH_SIZE" "WRITE_SIZE")
summ() {
python - "$1" <<'PY' >> "$1"
import sys
v = {}
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) >= 2 and p[0].isupper():
        try: v[p[0]] = float(p[1])
        except ValueError: pass
if "GRBM_GUI_ACTIVE" in v and v.get("SQ_WAVE_CYCLES"):
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    print(f"shader_cycles_per_launch {cyc:.4g}")
    print(f"mfma_pipe_utilisation {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc):.4f}   (1024 SIMDs)")
    print(f"valu_per_mfma {v.get('SQ_INSTS_VALU', 0) / max(v.get('SQ_INSTS_MFMA', 1), 1):.2f}   salu_per_mfma {v.get('SQ_INSTS_SALU', 0) / max(v.get('SQ_INSTS_MFMA', 1), 1):.2f}")
    print(f"wave_time_split active/issue-stall/parked {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
if "FETCH_SIZE" in v:
    print(f"hbm_side_bytes_per_launch fetch (x2 gfx950 correction) {2 * v['FETCH_SIZE'] * 1024:.4g}  write {v.get('WRITE_SIZE', 0) * 1024:.4g}")
PY
}
for shape in 22 19; do
  rm -rf /tmp/pg_*
  i=0
  for set in "${SETS[@]}"; do i=$((i+1)); rocprofv3 --pmc $set --output-format csv -d /tmp/pg_$i -o p -- $LAB $shape 0x4 > /tmp/pg_$i.log 2>&1 || tail -3 /tmp/pg_$i.log; done
  f=$OUT/gemm8p_counters_shape$shape.txt
  $LAB $shape 0x4 | grep -E "==|cfg 1" > $f
  echo "# per-launch means over the k_gemm8p<2,5,4,2> (256 x 320 tile) launches of the lab run above" >> $f
  python $GRAFT_REPO_ROOT/tools/micro/pmc_agg.py k_gemm8pILi2ELi5E /tmp/pg_*/p_counter_collection.csv >> $f
  summ $f
done
rm -rf /tmp/pg_*
i=0
for set in "${SETS[@]}"; do i=$((i+1)); rocprofv3 --pmc $set --output-format csv -d /tmp/pg_$i -o p -- python $GRAFT_REPO_ROOT/tools/micro/pmc_gemm_dma.py > /tmp/pg_$i.log 2>&1 || tail -3 /tmp/pg_$i.log; done
f=$OUT/gemm_dma_counters.txt
echo "# k_gemm_dma<128,128> on lin 368640 x {320, 2560 GEGLU, 960} x 320: per-launch means over the 12 launches (tools/micro/pmc_gemm_dma.py)" > $f
python $GRAFT_REPO_ROOT/tools/micro/pmc_agg.py k_gemm_dmaILi128ELi128E /tmp/pg_*/p_counter_collection.csv >> $f
summ $f
tail -8 $OUT/gemm8p_counters_shape22.txt; tail -6 $f
# strip-resident K = 320 Linear (cfg 12): the GEGLU feed-forward 368640 x 2560 x 320 (shape 5 of tools/micro/lin_lab)
if [ -x $GRAFT_REPO_ROOT/tools/micro/bin/lin_lab ]; then
LL=$GRAFT_REPO_ROOT/tools/micro/bin/lin_lab
rm -rf /tmp/pg_*
i=0
for set in "${SETS[@]}"; do i=$((i+1)); rocprofv3 --pmc $set --output-format csv -d /tmp/pg_$i -o p -- $LL 5 > /tmp/pg_$i.log 2>&1 || tail -3 /tmp/pg_$i.log; done
f=$OUT/lin_strip_counters.txt
$LL 5 | grep -E "==|cfg 12|cfg  1:" > $f
echo "# per-launch means over the k_lin_strip<320,4,false> launches of the lab run above" >> $f
python $GRAFT_REPO_ROOT/tools/micro/pmc_agg.py k_lin_stripILi320E /tmp/pg_*/p_counter_collection.csv >> $f
summ $f
tail -8 $f
fi
