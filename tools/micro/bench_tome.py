"""Matching (score GEMM + on-the-fly reduction + threshold + maps) per call: general tile-epilogue kernel vs the strip-resident C = 320 kernel."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.lib import lib
L=lib(); H=torch.float16; I=torch.int32
def st(): return torch.cuda.current_stream().cuda_stream
for na,nb,C in [(43200,14400,320),(31680,31680,320),(32400,10800,320),(23760,23760,320),(17280,5760,320),(12672,12672,320),(8100,2700,640),(10800,3600,640),(7920,7920,640),(5940,5940,640)]:
    T=na+nb
    x=torch.randn(2,T,C,device='cuda').to(H); m=torch.empty_like(x)
    L.tcl_tome_normalize_f16(x,m,2*T,C,st())
    a=torch.arange(0,na,dtype=I,device='cuda'); b=torch.arange(na,T,dtype=I,device='cuda')
    ws=torch.zeros(L.tcl_tome_match_workspace_bytes(na),dtype=torch.uint8,device='cuda')
    r=na//2; mrg=torch.empty(na-r+nb,dtype=I,device='cuda'); unm=torch.empty(T,dtype=I,device='cuda')
    res=[]
    for aff in (0,1):
        f=(lambda: L.tcl_tome_match_affine_f16(m,T*C,2,C,a,na,b,nb,r,na,0,na,mrg,unm,ws,st())) if aff else (lambda: L.tcl_tome_match_f16(m,T*C,2,C,a,na,b,nb,r,mrg,unm,ws,st()))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/10)
    gf=2*2*na*nb*C/1e9
    print(f"na={na} nb={nb} C={C}: tile kernel {res[0]*1e3:8.1f} us ({gf/res[0]:.0f} TF/s incl. threshold+maps)   strip kernel {res[1]*1e3:8.1f} us ({gf/res[1]:.0f} TF/s)")
