import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from tc_light_amd import sd15
from tc_light_amd.vae import VAEEngine
sd = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
vae = VAEEngine(sd, 'cuda')
imgs = torch.rand(30, 3, 720, 960, device='cuda')
ref = None
for bs in (2, 2, 5, 10, 15):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        z = vae.encode_imgs_batch(imgs, bs)
        torch.cuda.synchronize(); t1 = time.time()
        y = vae.decode_latents_batch(z, bs)
        torch.cuda.synchronize(); t2 = time.time()
    if ref is None: ref = (z.clone(), y.clone())
    dz = ((z.float() - ref[0].float()).norm() / ref[0].float().norm()).item(); dy = ((y - ref[1]).norm() / ref[1].norm()).item()
    print(f"batch {bs:2d}: encode {1e3*(t1-t0)/30:.2f} ms/frame  decode {1e3*(t2-t1)/30:.2f} ms/frame   vs batch 2: dz {dz:.2e} dy {dy:.2e}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
