import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
ws=torch.empty(96<<20,dtype=torch.uint8,device='cuda'); L.tcl_set_workspace(ws,ws.numel())
out=[]
for B,Hh,Ww,Ci,Co in [(8,4,12,1280,1280),(8,8,23,1280,1280),(8,4,12,2560,1280),(8,15,45,640,640),(8,12,15,1280,1280)]:
    x=torch.randn(B,Hh,Ww,Ci,device='cuda').to(H); w=torch.randn(Co,9*Ci,device='cuda').to(H); y=torch.empty(B,Hh,Ww,Co,device='cuda',dtype=H)
    t=timeit(lambda: L.tcl_conv3x3_f16(x,w,0,0,y,B,Hh,Ww,Ci,Co,1,1,0,0,0,st()))
    out.append(f"c{Hh}x{Ww} {Ci}->{Co}: {t:6.1f}")
for M,N,K in [(1472,1280,1280),(1472,1280,5120),(384,1280,2560),(5400,640,2560),(1472,10240,1280)]:
    A=torch.randn(M,K,device='cuda').to(H); W=torch.randn(N,K,device='cuda').to(H); C=torch.empty(M,N,device='cuda',dtype=H)
    t=timeit(lambda: L.tcl_gemm_f16(A,W,0,0,C,M,N,K,K,K,N,N,0,st()))
    out.append(f"g{M}x{N}x{K}: {t:6.1f}")
print(os.environ.get("TCL_GEMM_XCDN"), " | ".join(out))
