"""Which GEMM / conv shapes one UNet call of the metric's configuration (4-frame chunk, 160x90 latents, VidToMe on) launches, how often, and
what each achieves with the committed tile table: calls, us per call, TFLOP/s, algorithmic GB/s, share of the GEMM time of the call."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import collections
import torch
from tc_light_amd import sd15
from tc_light_amd.unet import UNetEngine, Ops
from tc_light_amd.vidtome import VidToMe
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
calls = collections.Counter()
og, oc = Ops.gemm, Ops.conv3x3
def gemm(self, a, w, bias=None, resid=None, act=0, out=None, M=None, lda=None, N=None, K=None, ldw=None, ldc=None):
    n, k = (N, K) if N is not None else w.shape
    m = M if M is not None else a.numel() // k
    calls[('g', m, n, k, act, resid is not None)] += 1
    return og(self, a, w, bias, resid, act, out, M, lda, N, K, ldw, ldc)
def conv(self, x, B, Hh, Ww, cin, w, bias, resid=None, stride=1, pad=1, up=None):
    calls[('c', B, Hh, Ww, cin, w.shape[0], stride, up)] += 1
    return oc(self, x, B, Hh, Ww, cin, w, bias, resid, stride, pad, up)
Ops.gemm, Ops.conv3x3 = gemm, conv
sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
eng = UNetEngine(sd, 'cuda', VidToMe('cuda', seed=1))
F, Hh, Ww = 4, 90, 160
text = torch.randn(2, 154, 768, device='cuda').half()
for _ in range(2):        # second call: the global-token bank exists (steady-state token counts)
    calls.clear()
    eng.forward_nhwc(torch.randn(2 * F, Hh, Ww, 8, device='cuda').half(), F, Hh, Ww, 801.0, text)
Ops.gemm, Ops.conv3x3 = og, oc
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
rows = []
for k, c in calls.items():
    if k[0] == 'g':
        _, M, N, K, act, hasr = k
        A = torch.randn(M, K, device='cuda').to(H); W = torch.randn(N, K, device='cuda').to(H)
        No = N // 2 if act == 2 else N
        C = torch.empty(M, No, device='cuda', dtype=H); R = torch.randn(M, No, device='cuda').to(H) if hasr else 0
        us = timeit(lambda: L.tcl_gemm_f16(A, W, 0, R, C, M, N, K, K, K, No, No, act, st()))
        fl, by = 2.0 * M * N * K, 2.0 * (M * K + N * K + M * No * (2 if hasr else 1))
        name = f"linear M={M} N={N} K={K} act={act} resid={int(hasr)}"
    else:
        _, B, h, w, ci, co, stride, up = k
        x = torch.randn(B, h, w, ci, device='cuda').to(H); wt = torch.randn(co, 9 * ci, device='cuda').to(H)
        Hu, Wu = up if up else (h, w); Ho = (Hu - 1) // stride + 1; Wo = (Wu - 1) // stride + 1
        y = torch.empty(B, Ho, Wo, co, device='cuda', dtype=H)
        us = timeit(lambda: L.tcl_conv3x3_f16(x, wt, 0, 0, y, B, h, w, ci, co, stride, 1, up[0] if up else 0, up[1] if up else 0, 0, st()))
        fl, by = 2.0 * B * Ho * Wo * 9 * ci * co, 2.0 * (B * h * w * ci + 9 * ci * co + B * Ho * Wo * co)
        name = f"conv3x3 B={B} {h}x{w} {ci}->{co} s{stride}{' up' if up else ''}"
    rows.append((us * c, name, c, us, fl / us / 1e6, by / us / 1e3))
tot = sum(r[0] for r in rows)
print(f"GEMM / conv time of one UNet call (alone, table configuration): {tot / 1e3:.1f} ms in {sum(r[2] for r in rows)} launches")
for t, name, c, us, tf, gb in sorted(rows, reverse=True)[:40]:
    print(f"{100 * t / tot:5.1f} %  {c:3d} x {us:8.1f} us  {tf:7.1f} TFLOP/s {gb:7.0f} GB/s  {name}")
