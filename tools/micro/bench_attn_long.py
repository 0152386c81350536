"""Sustained head_dim-40 attention rate (200 / 600 back-to-back calls at T = 35 640: the clock settles after ~50 ms).  RAMP=x scales the keys by a
ramp 0.3 .. x along the sequence (row maxima that keep climbing: the speculative softmax's bad case); TCL_FLASH40=5 selects the exact kernel."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time, subprocess
from tc_light_amd.lib import lib
L = lib(); H = torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
d, B, Hh, T = 40, 2, 8, 35640
C = Hh * d
q, k, v = (torch.randn(B, T, C, device="cuda").to(H) for _ in range(3)); o = torch.empty_like(q)
if os.environ.get('RAMP'):
    k = (k.float() * torch.linspace(0.3, float(os.environ['RAMP']), T, device='cuda')[None, :, None]).to(H)
wq = torch.empty(L.tcl_attention_q_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda")
wkv = torch.empty(L.tcl_attention_kv_bytes(B, Hh, T, d), dtype=torch.uint8, device="cuda")
f = lambda: L.tcl_attention_f16(q, C, T * C, k, C, T * C, v, C, T * C, o, C, T * C, B, Hh, T, T, d, d ** -0.5, 1, 1, wq, wkv, st())
f(); torch.cuda.synchronize()
for n in (200, 600):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / n
    smi = subprocess.run("rocm-smi --showpower --showclocks | grep -i 'sclk\\|Power (W)' | tr '\\n' ' '", shell=True, capture_output=True, text=True).stdout
    print(f"n={n:5d}: {ms*1e3:8.1f} us {4.0*B*Hh*T*T*d/ms/1e9:7.1f} TF/s")
