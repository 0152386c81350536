"""rel-L2 of the head_dim-40 attention against an f32 reference on heavy-tailed logits (q scaled) and on rows with a sink key;
TCL_FLASH40=5 selects the exact-maximum kernel for comparison."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
d, B, Hh, T = 40, 2, 8, 16400
C = Hh * d
g = torch.Generator(device="cuda").manual_seed(3)
for qs, outl in ((1.0, 0), (3.0, 0), (6.0, 0), (1.0, 14.0), (1.0, 18.0)):
    q = (torch.randn(B, T, C, device="cuda", generator=g) * qs).to(H)
    k = torch.randn(B, T, C, device="cuda", generator=g).to(H)
    v = torch.randn(B, T, C, device="cuda", generator=g).to(H)
    if outl:      # one outlier key per row block: key 5000 aligned with a common direction added to every query -> a sink `outl` nats above
        u = torch.randn(C, device="cuda", generator=g); 
        for h in range(Hh):
            uh = u[h*d:(h+1)*d]; uh /= uh.norm()
        q = (q.float() + u * (outl * (d ** 0.5)) ** 0.5).to(H); k[:, 5000] = (u * (outl * (d ** 0.5)) ** 0.5).to(H)
    o = torch.empty_like(q)
    wq = torch.empty(L.tcl_attention_q_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda")
    wkv = torch.empty(L.tcl_attention_kv_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda")
    L.tcl_attention_f16(q, C, T * C, k, C, T * C, v, C, T * C, o, C, T * C, B, Hh, T, T, d, d ** -0.5, 1, 1, wq, wkv, st())
    rows = torch.arange(0, T, 37, device="cuda")
    qq = q[:, rows].float().view(B, -1, Hh, d).transpose(1, 2)
    kk, vv = (t.float().view(B, T, Hh, d).transpose(1, 2) for t in (k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, -1, C)
    e = (o[:, rows].float() - ref)
    s = (qq @ kk.transpose(-1, -2)) * d ** -0.5 * 1.4427
    print(f"q scale {qs} outlier {outl}: logit std {s.std().item():.2f} bits, row max - median {(s.max(-1).values - s.median(-1).values).mean().item():.1f} bits; rel-L2 {(e.norm() / ref.norm()).item():.2e}, max abs {e.abs().max().item():.2e}")
