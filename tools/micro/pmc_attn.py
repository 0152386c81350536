import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16
def st(): return torch.cuda.current_stream().cuda_stream
d,B,Tq,Tk,kvd=40,2,35640,35640,1
Hh=8; C=Hh*d
q=torch.randn(B,Tq,C,device='cuda').to(H); k=torch.randn(B//kvd,Tk,C,device='cuda').to(H); v=torch.randn(B//kvd,Tk,C,device='cuda').to(H); o=torch.empty_like(q)
wq=torch.empty(L.tcl_attention_q_bytes(B,Hh,Tq,d),dtype=torch.uint8,device='cuda'); wkv=torch.empty(L.tcl_attention_kv_bytes(B//kvd,Hh,Tk,d),dtype=torch.uint8,device='cuda')
for _ in range(2): L.tcl_attention_f16(q,C,Tq*C,k,C,Tk*C,v,C,Tk*C,o,C,Tq*C,B,Hh,Tq,Tk,d,d**-0.5,kvd,1,wq,wkv,st())
torch.cuda.synchronize()
