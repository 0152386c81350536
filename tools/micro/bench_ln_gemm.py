"""norm -> Linear of the C = 320 transformer blocks: tcl_layernorm_f16 + tcl_gemm_f16 (table tile) against the fused tcl_ln_gemm_f16, us per call."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
from tc_light_amd.unet import Ops
L = lib(); H = torch.float16; ops = Ops(torch.device("cuda"))
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
C = 320
for M in (368640, 1474560):
    x = torch.randn(M, C, device="cuda").to(H); ga, be = torch.randn(C, device="cuda").to(H), torch.randn(C, device="cuda").to(H)
    y = torch.empty_like(x)
    for N, act in ((320, 0), (2560, 2)):
        W = (torch.randn(N, C, device="cuda") / 18).to(H); b = torch.randn(N, device="cuda").to(H)
        No = N // 2 if act == 2 else N
        c = torch.empty(M, No, device="cuda", dtype=H)
        t_ln = timeit(lambda: L.tcl_layernorm_f16(x, ga, be, y, M, C, 1e-5, st()))
        t_g = timeit(lambda: L.tcl_gemm_f16(y, W, b, 0, c, M, N, C, C, C, No, N, act, st()))
        t_f = timeit(lambda: L.tcl_ln_gemm_f16(x, ga, be, 1e-5, W, b, 0, c, M, N, C, C, C, No, N, act, st()))
        print(f"M={M} N={N} act={act}: layernorm {t_ln:.1f} + gemm {t_g:.1f} = {t_ln + t_g:.1f} us   fused {t_f:.1f} us")
