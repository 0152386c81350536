import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16
def st(): return torch.cuda.current_stream().cuda_stream
M=N=K=4096
A=torch.randn(M,K,device='cuda').to(H); W=torch.randn(N,K,device='cuda').to(H); C=torch.empty(M,N,device='cuda',dtype=H)
for _ in range(3): L.tcl_gemm_f16(A,W,0,0,C,M,N,K,K,K,N,N,0,st())
torch.cuda.synchronize()
B,Hh,Ww,Ci,Co=8,90,120,320,320
x=torch.randn(B,Hh,Ww,Ci,device='cuda').to(H); w=torch.randn(Co,9*Ci,device='cuda').to(H); y=torch.empty(B,Hh,Ww,Co,device='cuda',dtype=H)
for _ in range(3): L.tcl_conv3x3_f16(x,w,0,0,y,B,Hh,Ww,Ci,Co,1,1,0,0,0,st())
torch.cuda.synchronize()
