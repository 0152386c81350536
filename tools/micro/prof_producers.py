"""Round 6: the stage-2 input producers under the same discipline as the pass.  One script, run plain (wall clock per unit) or under
`rocprofv3 --kernel-trace --stats` (kernel table -> profiles/):  MemFlowNet on `--pairs` frame pairs at 1280x720 (15 GMA-SK2 iterations each, working
memory of 2 frames; reference inference_core_skflow.py:20-54), BriaRMBG on 8 frames, get_soft_mask_bwds + get_flowid (flow_utils.py:40-93) on the
metric's 300 x 1280 x 720 clip.  Prints one JSON line: wall ms per unit, the GEMM-class FLOPs the library's profiler counted and their bracket time."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tc_light_amd import flow_ids as FI
from tc_light_amd import memflow as MF
from tc_light_amd import rmbg as RM
from tc_light_amd.lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="memflow,rmbg,ids")
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--ids_frames", type=int, default=300)
ap.add_argument("--H", type=int, default=720)
ap.add_argument("--W", type=int, default=1280)
a = ap.parse_args()
dev = torch.device("cuda:0")
L = lib()
g = torch.Generator(device="cpu").manual_seed(12345)
base = torch.rand(1, 3, a.H // 8, a.W // 8, generator=g)
base = torch.nn.functional.interpolate(base, size=(a.H, a.W), mode="bilinear", align_corners=False)
fr = torch.cat([torch.roll(base, shifts=(k // 2, (3 * k) // 2), dims=(2, 3)) for k in range(a.frames)]).clamp(0, 1).to(dev)
res = {"H": a.H, "W": a.W}


def gemm_prof(fn):
    L.tcl_prof_begin(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
    L.tcl_prof_end(0, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt))
    return out, dt, ms.value, fl.value, cnt.value


if "memflow" in a.what:
    eng = MF.MemFlowEngine(MF.seeded_state_dict(MF.memflow_param_shapes(), 31), dev)
    MF.estimate_flows(eng, fr[:4], warm_start=False)                      # tiles of its shapes tuned / tabled, panels allocated, step graphs captured
    torch.cuda.synchronize(); t0 = time.perf_counter()
    MF.estimate_flows(eng, fr, warm_start=False)
    torch.cuda.synchronize(); dt_graph = time.perf_counter() - t0
    os.environ["TCL_MEMFLOW_GRAPH"] = "0"                                 # eager launches: what round 5 timed, and the only mode the launch profiler can bracket
    (fut, past), dt, ms, fl, cnt = gemm_prof(lambda: MF.estimate_flows(eng, fr, warm_start=False))
    os.environ.pop("TCL_MEMFLOW_GRAPH")
    pairs = 2 * (a.frames - 1)
    res["memflow"] = {"pairs": pairs, "ms_per_pair": dt_graph / pairs * 1e3, "ms_per_pair_eager_launches": dt / pairs * 1e3,
                      "gemm_class_tflop_per_pair": fl / pairs / 1e12, "gemm_class_launches_per_pair": cnt / pairs,
                      "gemm_class_bracket_ms_per_pair": ms / pairs, "iters": eng.iters}
if "rmbg" in a.what:
    rm = RM.RMBGEngine(RM.random_state_dict(1), dev)
    rm.estimate_alpha(fr[:2])
    _, dt, ms, fl, cnt = gemm_prof(lambda: rm.estimate_alpha(fr))
    res["rmbg"] = {"frames": a.frames, "ms_per_frame": dt / a.frames * 1e3, "gemm_class_tflop_per_frame": fl / a.frames / 1e12,
                   "gemm_class_launches_per_frame": cnt / a.frames, "gemm_class_bracket_ms_per_frame": ms / a.frames}
if "ids" in a.what:
    n = a.ids_frames
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
    import synth
    d = synth.video_clip(n, a.H, a.W, seed=12345)
    frames, pf = d["frames"].to(dev), d["past_flows"].to(dev)
    ff = torch.zeros_like(pf)
    ff[:-1] = -pf[1:]                                                      # future flow of frame i ~ -(past flow of frame i + 1): a consistent synthetic pair
    FI.soft_masks_and_ids(frames[:4], ff[:4], pf[:4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    masks = FI.get_soft_mask_bwds(frames, ff, pf, alpha=0.5)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ids, k = FI.get_flowid(frames, ff, masks)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    P = a.H * a.W
    # algorithmic bytes: masks read img (12) + two flows (16) + warped taps ~ (12 + 8) and write 4 per pixel; ids: read frame + previous frame (24), flow 8, mask 4,
    # previous ids 4 (gather), write ids 4 per pixel
    res["ids"] = {"frames": n, "K": k, "soft_mask_ms": (t1 - t0) * 1e3, "flowid_ms": (t2 - t1) * 1e3,
                  "soft_mask_GBps_algorithmic": n * P * 52 / (t1 - t0) / 1e9, "flowid_GBps_algorithmic": n * P * 44 / (t2 - t1) / 1e9,
                  "note": "flowid is a sequential scan over frames (frame i's ids depend on frame i-1's): one launch chain per frame"}
print(json.dumps(res))
