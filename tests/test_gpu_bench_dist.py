"""GPU: `python bench.py --gpus 2` with NO launcher around it must start two ranks by itself and print ONE JSON line with "n_gpus": 2
(it silently ran one rank).  Two ranks share the test box's single GPU through gloo (TCL_DIST_BACKEND=gloo; on a node the
backend is RCCL, one GPU per rank) -- the whole N > 1 bench path executes: self-launch under torch.distributed.run, frame sharding, the yt-plane
all-gather of x and of the owned noise pieces per step, the decoded-frame all-gather under the decode, stage 1 dealt / stage 2 replicated, max-over-ranks timing, the profiled pass on every rank,
per-rank phase seconds and collective seconds / bytes in the line.  The reference's only multi-GPU device is scripts/relight.sh:17-33
(independent videos per GPU); SURVEY 8(e) is this engine's design."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_launches_two_ranks_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TCL_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "16", "--height", "256", "--width", "256", "--steps", "2",
           "--warmup", "0", "--epochs", "2", "--epochs_exposure", "2", "--no_cpu_baseline", "--no_extras", "--profile_steps", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                        # rank 0 prints the one line, nobody else prints JSON
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["value"] > 0 and r["scaling"] == "strong"
    pr = r["per_rank"]
    assert pr["frames"] == [8, 8]
    assert set(pr["phase_seconds_min_max"]) >= {"encode", "denoise", "decode", "stage1", "stage2", "total"}
    c = pr["collectives"]
    assert c["all_gather_frames"]["calls"] >= 2 + 1                 # x per step + concat_conds once
    assert c["all_gather_decoded_async"]["calls"] == 1              # the decoded frames: one 8-frame slab per rank, handed over while ... nothing is left to decode
    assert c["all_gather_decoded_async"]["bytes_per_rank"] == 2 * 8 * 3 * 256 * 256 * 4
    assert c["all_gather_yt_noise"]["calls"] == 2                   # one per denoising step: every rank's OWNED (frames x columns) pieces
    full = 16 * 4 * 32 * 32 * 2                                     # the clip's yt noise, f16: what an all-gather of disjoint pieces delivers (+ padding of uneven deals)
    assert full <= c["all_gather_yt_noise"]["bytes_per_rank"] // 2 <= 1.5 * full
    assert "all_reduce_yt_noise" not in c                           # rounds 2-4's zero-filled full-size all-reduce is gone
    assert c["all_reduce"]["calls"] == 2 + 1                        # stage 1 dealt over the ranks: the [N,3,4] gradient of its 2 iterations + the losses once
    assert pr["unet_flop_balance_max_over_mean"] >= 1.0 and len(pr["unet_executed_tflop"]) == 2
    assert pr["seconds_in_collectives_max"] > 0 and pr["collective_bytes_per_denoise_step_per_rank"] > 0
    assert r["roofline"]["launches"] > 0 and r["roofline_gemm"]["calls"] > 0 and r["roofline_match"]["calls"] > 0
    print("bench --gpus 2 (gloo, shared GPU):", {k: r[k] for k in ("value", "n_gpus", "ms_per_step")}, pr["phase_seconds_min_max"], c)
