#!/bin/bash
# GPU box (via gpurun): round-4 counter evidence -> gpurun_out/profiles_r4/
#   1. SQ counters + HBM-side traffic of the 8-phase GEMM / conv kernel k_gemm8q<4,2,2,5> (256 x 320) on conv 640->320 @90x160 x64 (lab shape 22) and
#      lin 368640x640x2560 (shape 19), and of k_gemm8q<2,4,4,2> (256 x 256) on conv 1280->1280 @23x40 x64 (shape 25): stand-alone lab, q kernels only
#   2. HBM-side bytes per head_dim-40 attention call in the pass (two --pmc passes over 2 denoising steps of 60 frames)
#   3. HBM-side bytes per stage-2 iteration (tools/collect_path2_traffic.sh)
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r4
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
    print(f"lds_bank_conflict_cycles_per_lds_active {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.4f}")
    print(f"wave_time_split active/issue-stall/parked {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
if "FETCH_SIZE" in v:
    print(f"hbm_side_bytes_per_launch fetch (x2 gfx950 correction) {2 * v['FETCH_SIZE'] * 1024:.4g}  write {v.get('WRITE_SIZE', 0) * 1024:.4g}")
PY
}
for spec in "22 k_gemm8qILi4ELi2ELi2ELi5E q2" "19 k_gemm8qILi4ELi2ELi2ELi5E q2" "25 k_gemm8qILi2ELi4ELi4ELi2E q1"; do
  set -- $spec
  rm -rf /tmp/pg_*
  i=0
  for cs in "${SETS[@]}"; do i=$((i+1)); rocprofv3 --pmc $cs --output-format csv -d /tmp/pg_$i -o p -- $LAB $1 0x0 > /tmp/pg_$i.log 2>&1 || tail -3 /tmp/pg_$i.log; done
  f=$OUT/gemm8q_counters_shape$1.txt
  $LAB $1 0x0 | grep -E "==|$3 " > $f
  echo "# per-launch means over the $2 launches of the lab run above (21 launches: 1 check + 5 x 4 timed)" >> $f
  python $GRAFT_REPO_ROOT/tools/micro/pmc_agg.py $2 /tmp/pg_*/p_counter_collection.csv >> $f
  summ $f
  tail -7 $f
done
rm -rf /tmp/pg_*
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o pm -- python $GRAFT_REPO_ROOT/bench.py --frames 60 --steps 2 --warmup 0 --no_cpu_baseline --no_extras --epochs 0 --epochs_exposure 1 --profile_steps 0 > /tmp/pm_$c.log 2>&1 || tail -3 /tmp/pm_$c.log
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pm_FETCH_SIZE/pm_counter_collection.csv /tmp/pm_WRITE_SIZE/pm_counter_collection.csv k_flashILi40 k_flashILi40ELi48ELi64ELi2ELi4ELi2ELi0ELi0E > $OUT/flash40_traffic.json
cat $OUT/flash40_traffic.json
bash $GRAFT_REPO_ROOT/tools/collect_path2_traffic.sh r4
