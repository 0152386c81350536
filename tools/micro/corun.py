"""Do two of the pass's heavy kernels gain from running side by side?  Stream A loops the head_dim-40 flash call, stream B a Linear / conv GEMM or
the VidToMe match; time for both loops alone and together (same iteration counts).  gain = (tA + tB) / t_together.
CORUN_A=match puts the match call on stream A (is the match a better neighbour of the store-bound Linears than of the flash kernel?)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L = lib(); H = torch.float16; I = torch.int32
dev = "cuda"
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def flash_fn(T=35640):
    d, B, Hh = 40, 2, 8; C = Hh * d
    q, k, v = (torch.randn(B, T, C, device=dev).to(H) for _ in range(3)); o = torch.empty_like(q)
    wq = torch.empty(L.tcl_attention_q_bytes(B, Hh, T, d), dtype=torch.uint8, device=dev)
    wkv = torch.empty(L.tcl_attention_kv_bytes(B, Hh, T, d), dtype=torch.uint8, device=dev)
    return lambda st: L.tcl_attention_f16(q, C, T * C, k, C, T * C, v, C, T * C, o, C, T * C, B, Hh, T, T, d, d ** -0.5, 1, 1, wq, wkv, st)
def gemm_fn(M, N, K, act=0):
    A = torch.randn(M, K, device=dev).to(H); W = torch.randn(N, K, device=dev).to(H); No = N // 2 if act == 2 else N
    Cc = torch.empty(M, No, device=dev, dtype=H)
    return lambda st: L.tcl_gemm_f16(A, W, 0, 0, Cc, M, N, K, K, K, No, No, act, st)
def conv_fn(B, h, w, ci, co):
    x = torch.randn(B, h, w, ci, device=dev).to(H); wt = torch.randn(co, 9 * ci, device=dev).to(H); y = torch.empty(B, h, w, co, device=dev, dtype=H)
    return lambda st: L.tcl_conv3x3_f16(x, wt, 0, 0, y, B, h, w, ci, co, 1, 1, 0, 0, 0, st)
def match_fn(na=43200, nb=14400, C=320):
    T = na + nb
    x = torch.randn(2, T, C, device=dev).to(H); m = torch.empty_like(x)
    L.tcl_tome_normalize_f16(x, m, 2 * T, C, torch.cuda.current_stream().cuda_stream)
    a = torch.arange(0, na, dtype=I, device=dev); b = torch.arange(na, T, dtype=I, device=dev)
    ws = torch.zeros(L.tcl_tome_match_workspace_bytes(na), dtype=torch.uint8, device=dev)
    r = na // 2; mrg = torch.empty(na - r + nb, dtype=I, device=dev); unm = torch.empty(T, dtype=I, device=dev)
    return lambda st: L.tcl_tome_match_affine_f16(m, T * C, 2, C, a, na, b, nb, r, na, 0, na, mrg, unm, ws, st)
def loop(fn, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n): fn(stream.cuda_stream)
def timed(jobs):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in {j[1] for j in jobs}: s.wait_event(e0)
    for fn, s, n in jobs: loop(fn, s, n)
    for s in {j[1] for j in jobs}: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
fl = match_fn() if os.environ.get("CORUN_A", "flash") == "match" else flash_fn()     # CORUN_A=match: the match on stream A instead of the flash call
nameA = os.environ.get("CORUN_A", "flash")
partners = [("linear 115200x2560x320 GEGLU", gemm_fn(115200, 2560, 320, 2)), ("linear 115200x320x1280", gemm_fn(115200, 320, 1280)),
            ("conv3x3 8x90x160 320->320", conv_fn(8, 90, 160, 320, 320)), ("conv3x3 8x23x40 1280->1280", conv_fn(8, 23, 40, 1280, 1280)),
            ("match 43200x14400", match_fn()), ("flash (second copy)", flash_fn())]
for name, fn in partners:
    for f, s in ((fl, sA), (fn, sB)): loop(f, s, 3)
    tA1 = timed([(fl, sA, 1)]); tB1 = timed([(fn, sB, 1)])
    nA, nB = 40, max(1, int(40 * tA1 / tB1))            # equal time on both sides when alone
    tA, tB = timed([(fl, sA, nA)]), timed([(fn, sB, nB)])
    tAB = timed([(fl, sA, nA), (fn, sB, nB)])
    print(f"{nameA} x{nA} {tA:7.1f} ms | {name} x{nB} {tB:7.1f} ms | together {tAB:7.1f} ms -> gain {(tA + tB) / tAB:5.3f}")
