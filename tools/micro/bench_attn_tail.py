"""head_dim-40 self-attention over a sweep of token counts: what the partial last round of 256-query blocks costs (TCL_FLASH_TAIL=0|1)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
d, B, Hh = 40, 2, 8
C = Hh * d
for T in [int(a) for a in sys.argv[1:]] or [35640, 32768, 40960, 30000, 26000, 21600, 17820, 16384]:
    q, k, v = (torch.randn(B, T, C, device="cuda").to(H) for _ in range(3)); o = torch.empty_like(q)
    wq = torch.empty(L.tcl_attention_q_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda")
    wkv = torch.empty(L.tcl_attention_kv_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: L.tcl_attention_f16(q, C, T * C, k, C, T * C, v, C, T * C, o, C, T * C, B, Hh, T, T, d, d ** -0.5, 1, 1, wq, wkv, st()))
    nb = B * Hh * ((T + 255) // 256)
    print(f"T={T:6d} blocks256={nb:5d} rounds={nb / 512:5.2f}: {ms * 1e3:9.1f} us  {4.0 * B * Hh * T * T * d / ms / 1e9:7.1f} TF/s (incl. pack)")
