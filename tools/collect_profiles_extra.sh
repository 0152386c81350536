#!/bin/bash
# Run on the GPU box (via gpurun): counter / traffic evidence for the kernels beside the flash kernel -> gpurun_out/profiles_<round>/
#   matching (k_tome_match320), GEMM / implicit-conv family (k_gemm8, k_gemm_dma), path 2 (stage-1 / stage-2 iteration kernels)
R=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
sq_summary() {   # file -> append derived figures (MFMA pipe utilisation etc.)
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
    print(f"valu_active_fraction {4 * v.get('SQ_ACTIVE_INST_VALU', 0) / (1024 * cyc):.4f}   (quad-cycle counter x4)")
    print(f"valu_per_mfma {v.get('SQ_INSTS_VALU', 0) / max(v.get('SQ_INSTS_MFMA', 1), 1):.2f}")
    print(f"wave_time_split active/issue-stall/parked {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} / {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
PY
}
bash $GRAFT_REPO_ROOT/tools/micro/pmc_run.sh tools/micro/pmc_tome.py k_tome_match320 > $OUT/tome_match320_sq_counters.txt 2>&1; sq_summary $OUT/tome_match320_sq_counters.txt
rm -rf /tmp/pmcq_*
bash $GRAFT_REPO_ROOT/tools/micro/pmc_run.sh tools/micro/pmc_gemm8.py k_gemm8 > $OUT/gemm8_sq_counters.txt 2>&1; sq_summary $OUT/gemm8_sq_counters.txt
rm -rf /tmp/pmcq_*
# path 2: kernel times of 20 iterations of each stage at config 2 and config 3 sizes, then HBM-side traffic per kernel (separate passes)
python $GRAFT_REPO_ROOT/tools/micro/bench_p2.py 30 720 960 20 > $OUT/path2_iteration.txt 2>&1
python $GRAFT_REPO_ROOT/tools/micro/bench_p2.py 300 720 1280 20 >> $OUT/path2_iteration.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/ktp2 -o kt -- python $GRAFT_REPO_ROOT/tools/micro/bench_p2.py 300 720 1280 20 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/ktp2 24 > $OUT/path2_kernel_stats_300x1280x720.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pp_$c -o pm -- python $GRAFT_REPO_ROOT/tools/micro/bench_p2.py 300 720 1280 6 > /dev/null 2>&1
done
python - > $OUT/path2_traffic_300x1280x720.txt <<'PY'
import csv, collections
f = collections.defaultdict(float); n = collections.Counter(); w = collections.defaultdict(float)
for r in csv.DictReader(open("/tmp/pp_FETCH_SIZE/pm_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE": f[r["Kernel_Name"][:60]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:60]] += 1
for r in csv.DictReader(open("/tmp/pp_WRITE_SIZE/pm_counter_collection.csv")):
    if r["Counter_Name"] == "WRITE_SIZE": w[r["Kernel_Name"][:60]] += float(r["Counter_Value"])
print("# HBM-side KiB counters per launch (FETCH_SIZE x2 = gfx950 correction for 128-B requests, MI355X_MICROARCH.md), 300 x 1280 x 720, 6 iterations of each stage")
print(f"{'kernel':62s} {'launches':>8s} {'fetch_MB_x2':>12s} {'write_MB':>10s}")
for k in sorted(f, key=lambda k: -(2 * f[k] + w[k])):
    if n[k]: print(f"{k:62s} {n[k]:8d} {2 * f[k] * 1024 / n[k] / 1e6:12.2f} {w[k] * 1024 / n[k] / 1e6:10.2f}")
PY
ls -la $OUT
