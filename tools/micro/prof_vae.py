"""VAE encode + decode of 16 frames at 1280x720 (the metric's frame size), batch 8: run under `rocprofv3 --kernel-trace --stats` for the per-kernel
table (tools/prof_summary.py); prints ms per frame."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from tc_light_amd import sd15
from tc_light_amd.vae import VAEEngine
sd = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
vae = VAEEngine(sd, 'cuda')
imgs = torch.rand(16, 3, 720, 1280, device='cuda')
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    z = vae.encode_imgs_batch(imgs, bs)
    torch.cuda.synchronize(); t1 = time.time()
    y = vae.decode_latents_batch(z, bs)
    torch.cuda.synchronize(); t2 = time.time()
print(f"batch {bs}: encode {1e3*(t1-t0)/16:.2f} ms/frame  decode {1e3*(t2-t1)/16:.2f} ms/frame")
