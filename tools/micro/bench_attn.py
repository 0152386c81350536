import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(fn,n=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for d,B,Tq,Tk,kvd in [(40,2,35640,35640,1),(40,2,8910,8910,1),(40,8,10800,154,4),(80,2,8910,8910,1),(80,8,2700,154,4),(160,8,690,690,1)]:
    Hh=8; C=Hh*d
    q=torch.randn(B,Tq,C,device='cuda').to(H); k=torch.randn(B//kvd,Tk,C,device='cuda').to(H); v=torch.randn(B//kvd,Tk,C,device='cuda').to(H); o=torch.empty_like(q)
    wq=torch.empty(L.tcl_attention_q_bytes(B,Hh,Tq,d),dtype=torch.uint8,device='cuda'); wkv=torch.empty(L.tcl_attention_kv_bytes(B//kvd,Hh,Tk,d),dtype=torch.uint8,device='cuda')
    ms=timeit(lambda: L.tcl_attention_f16(q,C,Tq*C,k,C,Tk*C,v,C,Tk*C,o,C,Tq*C,B,Hh,Tq,Tk,d,d**-0.5,kvd,1,wq,wkv,st()))
    print(f"d={d} B={B} Tq={Tq} Tk={Tk}: {ms*1e3:9.1f} us  {4.0*B*Hh*Tq*Tk*d/ms/1e9:7.1f} TF/s (incl. pack)")
