#!/bin/bash
# GPU box (via gpurun): the counter half of the round-5 evidence (the kernel-trace half is tools/collect_profiles.sh r5; in round 5 the first --pmc pass
# behind the 3e6-record kernel trace segfaulted inside rocprofv3 once, so the counter passes run as a call of their own) -> gpurun_out/profiles_r5/
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do     # separate counter passes, 2 denoising steps of the metric's workload (60 of the 300 frames keep them short)
  rm -rf /tmp/pm_$c
  for try in 1 2; do
    # TCL_TOME_STREAM=0: rocprofv3's counter mode crashed in round 5 whenever the side stream launched k_tome_match320 (segfault inside the tool's dispatch
    # callback, four tries out of five); counter collection serialises the kernels anyway, so the flash kernel's bytes per call do not depend on the stream
    TCL_TOME_STREAM=0 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o pm -- python $GRAFT_REPO_ROOT/bench.py --frames 60 --steps 2 --warmup 0 --no_cpu_baseline --no_extras --epochs 0 --epochs_exposure 1 --profile_steps 0 > /tmp/pm_$c.log 2>&1 && break
    echo "pass $c try $try failed"; tail -3 /tmp/pm_$c.log
  done
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pm_FETCH_SIZE/pm_counter_collection.csv /tmp/pm_WRITE_SIZE/pm_counter_collection.csv k_flashILi40 k_flashILi40ELi48ELi64ELi2ELi4ELi2ELi0ELi0E > $OUT/flash40_traffic.json
cat $OUT/flash40_traffic.json
[ "$1" = traffic_only ] && exit 0
bash $GRAFT_REPO_ROOT/tools/micro/pmc_run.sh tools/micro/pmc_attn.py k_flashILi40ELi48ELi64ELi2ELi4ELi2ELi0ELi1E > $OUT/flash40_sq_counters.txt 2>&1
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
print(f"valu_per_mfma {v.get('SQ_INSTS_VALU', 0) / max(v.get('SQ_INSTS_MFMA', 1), 1):.2f}")
print(f"wave_time_split active/issue-stall/parked {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
PY
tail -12 $OUT/flash40_sq_counters.txt
