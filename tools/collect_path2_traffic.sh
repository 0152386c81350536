#!/bin/bash
# Run on the GPU box (via gpurun): HBM-side bytes per stage-2 iteration of the metric's clip size (300 x 1280 x 720) in the bench's codebook regime
# (reuse 0.02: K ~ N H W, lazy Adam) and at realistic track lengths (reuse 0.7: K ~ 8.4e7, dense Adam) -> gpurun_out/profiles_<round>/path2_traffic.json
# Two separate rocprofv3 --pmc passes per regime (FETCH_SIZE, WRITE_SIZE), stage 2 only; units / gfx950 correction per MI355X_MICROARCH.md:
# the counters are KiB, FETCH_SIZE tallies 128-B requests at 64 B (x2), WRITE_SIZE as is.
R=${1:-r4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IT=${2:-38}      # two epochs of the 300-frame clip (19 iterations each): the second epoch's visits are re-visits (lazy Adam: run-ahead rows, round 6)
for reuse in 0.02 0.7; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pq_${c}_$reuse
    P2_STAGES=2 rocprofv3 --pmc $c --output-format csv -d /tmp/pq_${c}_$reuse -o pm -- python $GRAFT_REPO_ROOT/tools/micro/bench_p2.py 300 720 1280 $IT $reuse > /tmp/pq_${c}_$reuse.log 2>&1
  done
done
python - $IT > $OUT/path2_traffic.json <<'PY'
import csv, json, sys, collections
it = int(sys.argv[1])
out = {}
for reuse, key in (("0.02", "bench_codebook_regime"), ("0.7", "realistic_codebook_regime")):
    f = collections.defaultdict(float); w = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f"/tmp/pq_FETCH_SIZE_{reuse}/pm_counter_collection.csv")):
        if r["Counter_Name"] == "FETCH_SIZE": f[r["Kernel_Name"][:48]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:48]] += 1
    for r in csv.DictReader(open(f"/tmp/pq_WRITE_SIZE_{reuse}/pm_counter_collection.csv")):
        if r["Counter_Name"] == "WRITE_SIZE": w[r["Kernel_Name"][:48]] += float(r["Counter_Value"])
    # the script runs the stage twice (one warm call, one timed call): 2 x it iterations.  Per-iteration kernels = the ones tcl_unique_tensor_opt
    # launches inside its loop (path2.hip); its one-off kernels (scatter-mean init, ids check, final catch-up of all rows, final gather) and the
    # script's own input synthesis (torch kernels) are listed apart and NOT part of the per-iteration figure
    LOOP = ("k_adam_touched_frame", "k_adam_catchup_frame", "k_adam_step_rows_lazy", "k_gather_codebook", "k_codebook_bwd", "k_flow_loss", "k_pixel_losses",
            "k_ssim_fwd", "k_ssim_bwd", "k_pool2", "k_msssim_finalize", "k_loss_finalize", "k_adam(", "__amd_rocclr_fillBufferAligned")
    is_loop = lambda k: any(t in k for t in LOOP)
    byts = lambda k: (2 * f[k] + w[k]) * 1024
    lazy = any("k_gather_codebook_lazy" in k for k in f)
    # the plain gather: per iteration 2 b = 32 frames (dense schedule; the lazy schedule gathers with k_gather_codebook_lazy), plus ONE final gather of all
    # 300 frames per call (a one-off, 300 / 32 iteration-gathers worth of bytes)
    def fix(k):
        if k.startswith("k_gather_codebook("):
            return 0.0 if lazy else (2 * it) / (2 * it + 2 * 300 / 32)
        return 1.0
    tot = sum(byts(k) * fix(k) for k in f if is_loop(k)) / (2 * it)
    once = sum(byts(k) for k in f if not is_loop(k) and k.startswith(("k_", "void k_"))) / 2
    per_kernel = {k: {"launches": n[k], "MB_per_iteration": byts(k) * fix(k) / (2 * it) / 1e6} for k in sorted(f, key=lambda k: -byts(k)) if is_loop(k)}
    line = [l for l in open(f"/tmp/pq_FETCH_SIZE_{reuse}.log") if l.startswith("stage 2")]
    out[key] = {"hbm_bytes_per_stage2_iteration": tot, "one_off_bytes_per_stage": once, "iterations_per_call": it,
                "microbench_line_under_counters": line[-1].strip() if line else None, "per_kernel": per_kernel}
out["hbm_bytes_per_stage2_iteration"] = out["bench_codebook_regime"]["hbm_bytes_per_stage2_iteration"]
import hashlib, os
out["path2_hip_sha256"] = hashlib.sha256(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tc_light_amd", "csrc", "path2.hip"), "rb").read()).hexdigest()      # bench.py reports the traffic basis only for THIS source
out["how"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/micro/bench_p2.py 300 720 1280 <iterations_per_call> <reuse>, stage 2 only; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over every kernel of the process / iterations"
print(json.dumps(out, indent=1))
PY
head -c 1500 $OUT/path2_traffic.json
