import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from tc_light_amd import sd15
from tc_light_amd.vae import VAEEngine
sd = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
vae = VAEEngine(sd, 'cuda')
imgs = torch.rand(8, 3, 720, 960, device='cuda')
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    z = vae.encode_imgs_batch(imgs)
    torch.cuda.synchronize(); t1 = time.time()
    y = vae.decode_latents_batch(z)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"encode {1e3*(t1-t0)/8:.2f} ms/frame  decode {1e3*(t2-t1)/8:.2f} ms/frame")
