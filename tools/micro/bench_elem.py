"""HBM-streaming kernels of the denoise pass at the metric's shapes (300 x 1280 x 720: level 0 = 14 400 tokens per frame, C = 320): us per launch
and algorithmic TB/s (bytes read + written once) -- LayerNorm, VidToMe normalise, GroupNorm (stats + apply), gathers, concat.  Anything far under
~4.5 TB/s (what a plain copy reaches) is a coalescing / occupancy problem of the kernel, not of the memory system."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
def report(name, us, nbytes):
    print(f"{name:64s} {us:9.1f} us  {nbytes / us / 1e6:6.2f} TB/s")
rnd = lambda *s: torch.randn(*s, device="cuda").to(H)
# ---- copy roof
for mb in (118, 472):
    a = torch.empty(mb << 19, dtype=H, device="cuda"); b = torch.empty_like(a)
    report(f"torch copy_ {mb} MB -> {mb} MB", timeit(lambda: b.copy_(a)), 2 * a.numel() * 2)
# ---- LayerNorm
for rows, C in ((368640, 320), (1474560, 320), (92160, 640), (23040, 1280)):
    x, g, b = rnd(rows, C), rnd(C), rnd(C); y = torch.empty_like(x)
    report(f"layernorm rows={rows} C={C}", timeit(lambda: L.tcl_layernorm_f16(x, g, b, y, rows, C, 1e-5, st())), 2 * x.numel() * 2)
# ---- VidToMe normalise
for rows, C in ((57600, 320), (126720, 320), (14400, 640), (31680, 640)):
    x = rnd(rows, C); y = torch.empty_like(x)
    report(f"tome_normalize rows={rows} C={C}", timeit(lambda: L.tcl_tome_normalize_f16(x, y, rows, C, st())), 2 * x.numel() * 2)
# ---- GroupNorm (stats + reduce + apply): x read twice, y written once
for B, HW, C1, C2, silu in ((64, 14400, 320, 0, 1), (64, 14400, 320, 320, 1), (64, 3600, 640, 0, 1), (64, 3600, 640, 640, 1), (64, 920, 1280, 1280, 1), (64, 5760, 320, 0, 0)):
    x1 = rnd(B * HW, C1); x2 = rnd(B * HW, C2) if C2 else None; C = C1 + C2
    g, b = rnd(C), rnd(C); y = torch.empty(B * HW, C, dtype=H, device="cuda")
    ws = torch.zeros(L.tcl_groupnorm_workspace_bytes(B, C), dtype=torch.uint8, device="cuda")
    report(f"groupnorm B={B} HW={HW} C={C1}+{C2} silu={silu}",
           timeit(lambda: L.tcl_groupnorm_f16(x1, C1, x2 if x2 is not None else 0, C2, g, b, y, B, HW, 32, 1e-5, silu, ws, st())), 3 * B * HW * C * 2)
# ---- gathers (merge / unmerge): 2 batch entries
for n, T, C in ((31680, 57600, 320), (47520, 63360, 320), (7920, 14400, 640)):
    src = rnd(2, T, C); mp = torch.randint(0, T, (n,), device="cuda", dtype=torch.int32); out = torch.empty(2, n, C, dtype=H, device="cuda")
    report(f"gather_rows n={n} of T={T} C={C} (x2)", timeit(lambda: L.tcl_gather_rows_f16(src, T * C, 0, 0, mp, out, n * C, 2, n, C, st())), 2 * 2 * n * C * 2)
for n, T, C in ((57600, 47520, 320), (14400, 11880, 640)):
    h = rnd(2, n, C); y = rnd(2, T, C); mp = torch.randint(0, T, (n,), device="cuda", dtype=torch.int32)
    report(f"gather_add_rows n={n} from T={T} C={C} (x2)", timeit(lambda: L.tcl_gather_add_rows_f16(h, n * C, y, T * C, mp, 2, n, C, st())), 3 * 2 * n * C * 2)
for rows, C1, C2 in ((921600, 320, 320), (230400, 640, 640)):
    x1, x2 = rnd(rows, C1), rnd(rows, C2); y = torch.empty(rows, C1 + C2, dtype=H, device="cuda")
    report(f"concat rows={rows} {C1}+{C2}", timeit(lambda: L.tcl_concat_channels_f16(x1, C1, x2, C2, y, rows, st())), 2 * rows * (C1 + C2) * 2)
