import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, faulthandler
faulthandler.enable()
from tc_light_amd.lib import lib
if len(sys.argv)>1 and sys.argv[1]=='noauto': lib().tcl_gemm_autotune(0)
from tc_light_amd import sd15
from tc_light_amd.unet import UNetEngine
from tc_light_amd.vidtome import VidToMe
sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
tome = VidToMe("cuda", seed=5); eng = UNetEngine(sd, "cuda", tome)
Hh, Ww, t = 16, 24, 801.0
Fs=[1,3,2]
g=np.random.default_rng(0)
text=torch.from_numpy(g.standard_normal((2,77,768)).astype(np.float32)).cuda().half()
xs=[torch.randn(2*F,Hh,Ww,8,device='cuda').half() for F in Fs]
draws=[(0,0.9),(2,0.3),(1,0.7)]
tome.reset_global_tokens(); tome.draws=list(draws)
for x,F in zip(xs,Fs):
    print("seq",F,flush=True); e=eng.forward_nhwc(x,F,Hh,Ww,t,text); torch.cuda.synchronize(); print(" ok",e.float().abs().mean().item(),flush=True)
tome.reset_global_tokens(); tome.draws=list(draws)
print("many",flush=True)
m=eng.forward_many(xs,Fs,Hh,Ww,t,text); torch.cuda.synchronize(); print("ok many",flush=True)
