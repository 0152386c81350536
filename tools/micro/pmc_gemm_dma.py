"""A few launches of the 4-wave LDS-DMA GEMM (k_gemm_dma<128,128>) on the short-K Linear shapes of the metric's pass, for rocprofv3 --pmc:
q2 / o2 / proj (M x 320 x 320), ff1 GEGLU (M x 2560 x 320) and the per-chunk QKV (M x 960 x 320).  M = 368640 (a quarter of a pass)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
L.tcl_gemm_autotune(0); L.tcl_gemm_tune(1, 1)          # cfg 1 = k_gemm_dma 128x128, no K split
M = 368640
for N, K, act in ((320, 320, 0), (2560, 320, 2), (960, 320, 0)):
    A = torch.randn(M, K, device="cuda").to(H); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(H)
    No = N // 2 if act == 2 else N
    C = torch.empty(M, No, device="cuda", dtype=H)
    for _ in range(4):
        L.tcl_gemm_f16(A, W, 0, 0, C, M, N, K, K, K, No, N, act, st())
    torch.cuda.synchronize()
