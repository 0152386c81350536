import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from tc_light_amd.rmbg import RMBGEngine, random_state_dict
e=RMBGEngine(random_state_dict(1),'cuda')
fr=torch.rand(30,3,720,960,device='cuda')
e.estimate_alpha(fr[:2]); torch.cuda.synchronize()
t0=time.perf_counter(); a=e.estimate_alpha(fr); torch.cuda.synchronize(); t=time.perf_counter()-t0
print(f"RMBG matte for 30 frames 960x720 (net input 896x1152 after the reference's transposed resize): {t*1e3:.0f} ms total, {t/30*1e3:.1f} ms/frame; alpha {tuple(a.shape)}")
