import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tc_light_amd import sd15
from tc_light_amd.unet import UNetEngine
from tc_light_amd.vidtome import VidToMe
from tc_light_amd.lib import lib
mode=sys.argv[1]
L=lib()
if mode=="noauto": L.tcl_gemm_autotune(0)
if mode.startswith("cfg"): L.tcl_gemm_tune(int(mode[3:]),1)
sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
tome = VidToMe("cuda", seed=5); eng = UNetEngine(sd, "cuda", tome)
Hh, Ww, t = 16, 24, 801.0
F=3
g=np.random.default_rng(0)
text=torch.from_numpy(g.standard_normal((2,77,768)).astype(np.float32)).cuda().half()
x=torch.randn(2*F,Hh,Ww,8,device='cuda').half()
tome.reset_global_tokens(); tome.draws=[(2,0.3)]
e=eng.forward_nhwc(x,F,Hh,Ww,t,text); torch.cuda.synchronize(); print(mode,"ok",e.float().abs().mean().item(),flush=True)
