"""GroupNorm (stats + reduce + apply) alone at the pass's shapes: us per call.  A/B of two builds: TCL_LIB_PATH."""
import torch, time, sys
sys.path.insert(0, ".")
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
for B, HW, C in [(50, 14400, 320), (50, 14400, 960), (50, 3600, 1280), (2, 921600, 128)]:
    x = torch.randn(B, HW, C, device="cuda").to(H); ga = torch.randn(C, device="cuda").to(H); be = torch.randn(C, device="cuda").to(H); y = torch.empty_like(x)
    ws = torch.zeros(int(L.tcl_groupnorm_workspace_bytes(B, C)), dtype=torch.uint8, device="cuda")
    f = lambda: L.tcl_groupnorm_f16(x, C, 0, 0, ga, be, y, B, HW, 32, 1e-5, 1, ws, st())
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize(); print((B, HW, C), round((time.perf_counter() - t0) / 30 * 1e6, 1), "us")
