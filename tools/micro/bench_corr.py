import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd.memflow import CorrBlock
B,D,H,W=1,256,90,160
f1=torch.randn(B,D,H,W,device='cuda'); f2=torch.randn(B,D,H,W,device='cuda')
ys,xs=torch.meshgrid(torch.arange(H,device='cuda').float(),torch.arange(W,device='cuda').float(),indexing='ij')
co=torch.stack([xs,ys])[None]+4*torch.randn(B,2,H,W,device='cuda')
cb=CorrBlock(f1,f2); cb(co); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(15): cb(co)
e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/15
print(f"corr lookup {H}x{W} D={D} 4 levels r=4: {t*1e3:.1f} us per call; algorithmic 4*100*D MACs/px = {2*4*100*D*H*W/1e9:.2f} GFLOP -> {2*4*100*D*H*W/t/1e9:.2f} TFLOP/s; out {4*81*H*W*4/1e6:.1f} MB")
# reference way: volume + pyramid + grid_sample
t0=torch.cuda.Event(enable_timing=True); t1=torch.cuda.Event(enable_timing=True); t0.record()
vol=(f1.view(B,D,-1).transpose(1,2)@f2.view(B,D,-1)).view(B*H*W,1,H,W)/16
t1.record(); torch.cuda.synchronize(); print(f"torch all-pairs volume alone: {t0.elapsed_time(t1):.2f} ms, {vol.numel()*4/1e6:.0f} MB")
