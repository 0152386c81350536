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
print("GEMM shapes (M,N,K)")
for M,N,K in [(86400,320,320),(86400,960,320),(86400,2560,320),(86400,320,1280),(71280,960,320),(21600,640,640),(21600,5120,640),(21600,640,2560),(5520,1280,1280),(5520,10240,1280),(8192,8192,8192)]:
    A=torch.randn(M,K,device='cuda').to(H); W=torch.randn(N,K,device='cuda').to(H); C=torch.empty(M,N,device='cuda',dtype=H); b=torch.randn(N,device='cuda').to(H)
    ms=timeit(lambda: L.tcl_gemm_f16(A,W,b,0,C,M,N,K,K,K,N,N,0,st()))
    ms2=timeit(lambda: torch.matmul(A,W.t()))
    print(f"{M:6d} {N:6d} {K:6d}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TF/s | torch(hipblaslt) {ms2*1e3:8.1f} us {2*M*N*K/ms2/1e9:7.1f} TF/s | bytes {(M*K+M*N)*2/ms/1e6:6.0f} GB/s")
print("conv3x3 (B,H,W,Cin,Cout)")
for B,Hh,Ww,Ci,Co in [(8,90,120,320,320),(8,45,60,640,640),(8,23,30,1280,1280),(8,45,60,1920,640),(8,90,120,960,320),(2,360,480,256,256),(2,720,960,128,128)]:
    x=torch.randn(B,Hh,Ww,Ci,device='cuda').to(H); w=torch.randn(Co,9*Ci,device='cuda').to(H); y=torch.empty(B,Hh,Ww,Co,device='cuda',dtype=H); b=torch.randn(Co,device='cuda').to(H)
    ms=timeit(lambda: L.tcl_conv3x3_f16(x,w,b,0,y,B,Hh,Ww,Ci,Co,1,1,0,0,0,st()))
    fl=2*B*Hh*Ww*9*Ci*Co
    print(f"{B} {Hh}x{Ww} {Ci}->{Co}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s")
print("groupnorm")
for B,HW,C in [(8,10800,320),(8,2700,640),(8,10800,960)]:
    x=torch.randn(B,HW,C,device='cuda').to(H); g=torch.ones(C,device='cuda',dtype=H); y=torch.empty_like(x)
    ws=torch.zeros(L.tcl_groupnorm_workspace_bytes(B,C),dtype=torch.uint8,device='cuda')
    ms=timeit(lambda: L.tcl_groupnorm_f16(x,C,0,0,g,g,y,B,HW,32,1e-5,1,ws,st()))
    print(f"GN {B}x{HW}x{C}: {ms*1e3:8.1f} us  {x.numel()*2*3/ms/1e6:6.0f} GB/s")
