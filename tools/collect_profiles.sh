#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench + PMC traffic of the dominant kernel -> gpurun_out/profiles_rNN/
set -e
R=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# round 2: the default bench IS the metric's configuration (300 frames 1280x720, --steps 20 --warmup 5); ~3e6 kernel records
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --profile_steps 0 > $OUT/bench_under_rocprof.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt 60 --timed-pass > $OUT/bench_kernel_stats.txt
rm -rf /tmp/kt
if [ "$2" = "exclusive" ]; then
# the same with the matching chain on the main stream: per-kernel durations of kernels that have the GPU to themselves
TCL_TOME_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/ktx -o kt -- python $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_extras --profile_steps 0 > $OUT/bench_under_rocprof_exclusive.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ktx 60 --timed-pass > $OUT/bench_kernel_stats_exclusive.txt
rm -rf /tmp/ktx
fi
for c in FETCH_SIZE WRITE_SIZE; do     # separate counter passes, 2 denoising steps of the same workload (60 of the 300 frames keep them short)
  rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o pm -- python $GRAFT_REPO_ROOT/bench.py --frames 60 --steps 2 --warmup 0 --no_cpu_baseline --no_extras --epochs 0 --epochs_exposure 1 --profile_steps 0 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pm_FETCH_SIZE/pm_counter_collection.csv /tmp/pm_WRITE_SIZE/pm_counter_collection.csv k_flashILi40 k_flashILi40ELi48ELi64ELi2ELi4ELi2ELi0ELi0E > $OUT/flash40_traffic.json
cat $OUT/flash40_traffic.json; tail -1 $OUT/bench_under_rocprof.json | cut -c1-300
# SQ counters of the head_dim-40 flash kernel alone (tools/micro/pmc_attn.py: B=2, H=8, T=35640, the merged xy-plane sequence of config 2):
# MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x shader cycles), shader cycles = GRBM_GUI_ACTIVE / 8 XCDs
bash $GRAFT_REPO_ROOT/tools/micro/pmc_run.sh tools/micro/pmc_attn.py k_flashILi40ELi48ELi64ELi2ELi4ELi2ELi0ELi1E > $OUT/flash40_sq_counters.txt 2>&1      # the speculative kernel (the gated exact one behind it only reads its flags)
python - "$OUT/flash40_sq_counters.txt" <<'PY' >> $OUT/flash40_sq_counters.txt
import sys
v = {}
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) >= 2 and p[0].isupper():
        try: v[p[0]] = float(p[1])
        except ValueError: pass
cyc = v["GRBM_GUI_ACTIVE"] / 8
print(f"shader_cycles_per_launch {cyc:.4g}")
print(f"mfma_pipe_utilisation {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.4f}   (1024 SIMDs)")
print(f"valu_active_fraction {4 * v['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.4f}   (quad-cycle counter x4)")
print(f"wave_time_split active/issue-stall/parked {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
PY
cat $OUT/flash40_sq_counters.txt | tail -5
