"""Every tile configuration on the big Linear shapes of a config-2 UNet pass (rows = all chunks of the pass): TF/s per cfg id
(1 dma128x128, 2 dma64x128, 3 dma128x64, 4 dma64x64, 11 dma256x128, 5-8 the 8-wave kernels), '-' = configuration not applicable."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(fn, n=5):
    try: fn()
    except RuntimeError: return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
L.tcl_gemm_autotune(0)
CFGS = [1, 2, 3, 4, 11, 5, 6, 7, 8]
print("shape (M,N,K,act,resid)".ljust(34) + "".join(f"{c:>7d}" for c in CFGS))
for M, N, K, act, res in [(648000, 2560, 320, 2, 0), (648000, 320, 320, 0, 1), (648000, 320, 1280, 0, 1), (648000, 960, 320, 0, 0), (162000, 5120, 640, 2, 0),
                          (162000, 640, 640, 0, 1), (162000, 640, 2560, 0, 1), (41400, 10240, 1280, 2, 0), (41400, 1280, 1280, 0, 1), (41400, 1280, 5120, 0, 1),
                          (71280, 960, 320, 0, 0), (17820, 960, 320, 0, 0), (17820, 320, 320, 0, 0)]:
    A = torch.randn(M, K, device="cuda").to(H); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(H)
    No = N // 2 if act == 2 else N
    C = torch.empty(M, No, device="cuda", dtype=H); R = torch.randn(M, No, device="cuda").to(H) if res else 0
    row = f"{(M, N, K, act, res)}".ljust(34)
    for cfg in CFGS:
        L.tcl_gemm_tune(cfg, 1)
        t = timeit(lambda: L.tcl_gemm_f16(A, W, 0, R, C, M, N, K, K, K, No, No, act, st()))
        row += f"{2.0 * M * N * K / t / 1e9:7.0f}" if t else "      -"
    print(row)
    del A, C, R
