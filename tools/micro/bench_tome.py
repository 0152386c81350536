import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16; I=torch.int32
def st(): return torch.cuda.current_stream().cuda_stream
for na,nb,C in [(23760,11880,320),(11880,23760,320),(8100,2700,320),(5940,2970,640),(17820,17820,320)]:
    T=na+nb
    x=torch.randn(2,T,C,device='cuda').to(H); m=torch.empty_like(x)
    L.tcl_tome_normalize_f16(x,m,2*T,C,st())
    a=torch.arange(0,na,dtype=I,device='cuda'); b=torch.arange(na,T,dtype=I,device='cuda')
    ws=torch.zeros(L.tcl_tome_match_workspace_bytes(na),dtype=torch.uint8,device='cuda')
    r=na//2; mrg=torch.empty(na-r+nb,dtype=I,device='cuda'); unm=torch.empty(T,dtype=I,device='cuda')
    for _ in range(3): L.tcl_tome_match_f16(m,T*C,2,C,a,na,b,nb,r,mrg,unm,ws,st())
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): L.tcl_tome_match_f16(m,T*C,2,C,a,na,b,nb,r,mrg,unm,ws,st())
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
    print(f"na={na} nb={nb} C={C}: {ms*1e3:8.1f} us total/call, score GEMM {2*2*na*nb*C/1e9:.0f} GF -> {2*2*na*nb*C/ms/1e9:.0f} TF/s if all in the match kernel")
