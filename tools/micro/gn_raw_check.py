import torch,time,sys
sys.path.insert(0,".")
from tc_light_amd.lib import lib
L=lib(); H=torch.float16
st=lambda: torch.cuda.current_stream().cuda_stream
for B,HW,C1,C2 in [(50,14400,320,320),(50,14400,640,320),(50,3600,1280,640)]:
    C=C1+C2
    x1=torch.randn(B,HW,C1,device="cuda").to(H); x2=torch.randn(B,HW,C2,device="cuda").to(H); ga=torch.randn(C,device="cuda").to(H); be=torch.randn(C,device="cuda").to(H); y=torch.empty(B,HW,C,device="cuda",dtype=H); yr=torch.empty_like(y)
    ws=torch.zeros(int(L.tcl_groupnorm_workspace_bytes(B,C)),dtype=torch.uint8,device="cuda")
    f=lambda: L.tcl_groupnorm_concat_f16(x1,C1,x2,C2,ga,be,y,yr,B,HW,32,1e-5,1,ws,st())
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print((B,HW,C1,C2), round((time.perf_counter()-t0)/20*1e6,1),"us", bool(torch.equal(yr, torch.cat([x1,x2],-1))))
