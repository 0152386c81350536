"""Stage 1 / stage 2 of path 2 alone, whole-stage drivers, at a BASELINE size: ms per iteration and the algorithmic HBM rate (SURVEY 8(d):
stage 2 = (56 + 48 + 24) b P + 84 K bytes per iteration, stage 1 = 2 x 60 b P).  usage: bench_p2.py [frames H W iters [reuse]]   (default: config 2)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch, time
from tc_light_amd import post_opt as P
n, h, w, iters = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (30, 720, 960, 20)))
reuse = float(sys.argv[5]) if len(sys.argv) > 5 else 0.7      # fraction of a frame's pixels that continue a track (0.02: K ~ N H W, the bench clip's regime)
g = torch.Generator(device="cuda").manual_seed(1)
base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, h + 8, w + n + 8, device="cuda", generator=g), 9, stride=1, padding=4)
ed = torch.stack([base[0, :, 4:4 + h, i:i + w] for i in range(n)]).contiguous()
ed = (ed * (1 + 0.03 * torch.randn(n, 3, 1, 1, device="cuda", generator=g)) + 0.01 * torch.randn(ed.shape, device="cuda", generator=g)).clamp_(0, 1)
# a smooth, spatially varying flow field (a few px, integer crossings every ~100 px like camera / object motion) + estimator noise
yy, xx = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
flows = torch.stack([1.3 + 1.5 * torch.sin(xx / 160 + yy / 300), 0.4 + 1.0 * torch.cos(yy / 120 - xx / 400)])[None].repeat(n, 1, 1, 1)
flows = flows * (1 + 0.1 * torch.randn(n, 1, 1, 1, device="cuda", generator=g)) + 0.02 * torch.randn(flows.shape, device="cuda", generator=g); flows[0] = 0
masks = (torch.rand(n, 1, h, w, device="cuda", generator=g) > 0.1).float()
ids = torch.empty(n, h, w, dtype=torch.int64, device="cuda"); ids[0] = torch.arange(h * w, device="cuda").view(h, w); last = h * w
for k in range(1, n):
    fresh = torch.rand(h, w, device="cuda", generator=g) > reuse; fresh[:, 0] = True
    cur = torch.roll(ids[k - 1], 1, dims=1); c = int(fresh.sum()); cur[fresh] = last + torch.arange(c, device="cuda"); last += c; ids[k] = cur
inv, K = ids.reshape(-1).to(torch.int32), last
rng = np.random.default_rng(0)
sched = P.make_schedule(n, 16, -(-iters // -(-n // 16)), rng)[:iters]
b, Pp = 16, h * w
for stage in [int(v) for v in os.environ.get('P2_STAGES', '1,2').split(',')]:        # P2_STAGES=2: stage 2 only (traffic passes)
    ds = P.OptDataset(ed, flows, masks, device="cuda")
    f = (lambda: P.exposure_align(ds, sched, epochs=1, batch_size=16)) if stage == 1 else (lambda: P.unique_tensor_optimization(ds, inv, sched, batch_size=16, k=K))
    f(); torch.cuda.synchronize(); ds = P.OptDataset(ed, flows, masks, device="cuda")
    ds.flow_shift; torch.cuda.synchronize()              # (once per clip: the per-frame fixed-point scale of the flow scatter, not an iteration's work)
    t0 = time.perf_counter(); res = f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(sched)
    if os.environ.get("P2_DIGEST"):          # A/B of two library builds (TCL_LIB_PATH): the stage's outputs must agree bit for bit
        import hashlib
        hs = [hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16] for t in (res if isinstance(res, (tuple, list)) else [res]) if torch.is_tensor(t)]
        print(f"stage {stage} digest: {' '.join(hs)}")
    by = (2 * 60 * b * Pp) if stage == 1 else ((56 + 48 + 24) * b * Pp + 84 * K)
    print(f"stage {stage}: {n} frames {w}x{h}, K={K}: {dt * 1e3:.3f} ms/iteration, algorithmic {by / 1e9:.2f} GB/iteration -> {by / dt / 1e12:.2f} TB/s ({by / dt / 8e12 * 100:.1f} % of 8 TB/s)")
