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
print("cfg", os.environ.get("TCL_GEMM_BIG"), os.environ.get("TCL_GEMM_DMA"))
for M,N,K in [(86400,320,320),(86400,960,320),(86400,2560,320),(86400,320,1280),(21600,640,640),(21600,5120,640),(21600,640,2560),(5520,1280,1280),(5520,10240,1280),(8192,8192,8192),(4096,4096,4096)]:
    A=torch.randn(M,K,device='cuda').to(H); W=torch.randn(N,K,device='cuda').to(H); C=torch.empty(M,N,device='cuda',dtype=H); b=torch.randn(N,device='cuda').to(H)
    ms=timeit(lambda: L.tcl_gemm_f16(A,W,b,0,C,M,N,K,K,K,N,N,0,st()))
    ref=(A[:512].float()@W.float().t()+b.float())
    err=((C[:512].float()-ref).norm()/ref.norm()).item()
    print(f"{M:6d} {N:6d} {K:6d}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TF/s  relerr {err:.2e}")
for B,Hh,Ww,Ci,Co in [(8,90,120,320,320),(8,45,60,640,640),(8,23,30,1280,1280),(8,90,120,960,320)]:
    x=torch.randn(B,Hh,Ww,Ci,device='cuda').to(H); w=torch.randn(Co,9*Ci,device='cuda').to(H); y=torch.empty(B,Hh,Ww,Co,device='cuda',dtype=H); b=torch.randn(Co,device='cuda').to(H)
    ms=timeit(lambda: L.tcl_conv3x3_f16(x,w,b,0,y,B,Hh,Ww,Ci,Co,1,1,0,0,0,st()))
    ref=torch.nn.functional.conv2d(x[:1].permute(0,3,1,2).float(), w.view(Co,3,3,Ci).permute(0,3,1,2).float(), b.float(), padding=1).permute(0,2,3,1)
    err=((y[:1].float()-ref).norm()/ref.norm()).item()
    print(f"conv {B} {Hh}x{Ww} {Ci}->{Co}: {ms*1e3:8.1f} us {2*B*Hh*Ww*9*Ci*Co/ms/1e9:7.1f} TF/s relerr {err:.2e}")
