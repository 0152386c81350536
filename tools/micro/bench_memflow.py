import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tc_light_amd import memflow as MF
sd=MF.seeded_state_dict(MF.memflow_param_shapes(), 31)
eng=MF.MemFlowEngine(sd,'cuda')
fr=torch.rand(4,3,720,960,device='cuda')*2-1
for i in range(2): eng.step(torch.stack([fr[i],fr[i+1]])[None])
torch.cuda.synchronize(); t0=time.perf_counter()
low,up=eng.step(torch.stack([fr[2],fr[3]])[None]); torch.cuda.synchronize(); t=time.perf_counter()-t0
print(f"MemFlowEngine.step 960x720 (90x120 grid, 15 iterations, 2-frame memory): {t*1e3:.0f} ms per frame pair", tuple(up.shape))
