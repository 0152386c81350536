import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16; I=torch.int32
def st(): return torch.cuda.current_stream().cuda_stream
na,nb,C=31680,31680,320      # config 3 two-set merge
T=na+nb
x=torch.randn(2,T,C,device='cuda').to(H); m=torch.empty_like(x)
L.tcl_tome_normalize_f16(x,m,2*T,C,st())
a=torch.arange(0,na,dtype=I,device='cuda'); b=torch.arange(na,T,dtype=I,device='cuda')
ws=torch.zeros(L.tcl_tome_match_workspace_bytes(na),dtype=torch.uint8,device='cuda')
r=na//2; mrg=torch.empty(na-r+nb,dtype=I,device='cuda'); unm=torch.empty(T,dtype=I,device='cuda')
for _ in range(2): L.tcl_tome_match_affine_f16(m,T*C,2,C,a,na,b,nb,r,na,0,na,mrg,unm,ws,st())
torch.cuda.synchronize()
