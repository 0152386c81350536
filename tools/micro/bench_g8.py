import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16
def st(): return torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
out=[]
for M,N,K in [(4096,4096,4096),(65536,1280,2880),(65536,320,2880)]:
    A=torch.randn(M,K,device='cuda').to(H); W=torch.randn(N,K,device='cuda').to(H); C=torch.empty(M,N,device='cuda',dtype=H)
    ms=timeit(lambda: L.tcl_gemm_f16(A,W,0,0,C,M,N,K,K,K,N,N,0,st()))
    out.append(f"{M}x{N}x{K}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:6.0f} TF/s")
for B,Hh,Ww,Ci,Co in [(8,64,128,320,320),(8,64,128,1280,1280)]:
    x=torch.randn(B,Hh,Ww,Ci,device='cuda').to(H); w=torch.randn(Co,9*Ci,device='cuda').to(H); y=torch.empty(B,Hh,Ww,Co,device='cuda',dtype=H)
    ms=timeit(lambda: L.tcl_conv3x3_f16(x,w,0,0,y,B,Hh,Ww,Ci,Co,1,1,0,0,0,st()))
    out.append(f"conv {Ci}->{Co}@{Hh}x{Ww}: {ms*1e3:7.1f} us {2*B*Hh*Ww*9*Ci*Co/ms/1e9:6.0f} TF/s")
print(os.environ.get("TCL_GEMM8"), " | ".join(out))
