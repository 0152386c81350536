"""Round 6: where the 9.86e-4 of the decoded frames comes from.  TEST-SIDE TOOL (imports oracle/).
configs[0]'s frame size (512 x 512), seeded weights.  Measures, each against the f32 oracle:
  * VAE encode alone (engine, and the oracle with f16 op outputs = the floor of an op-by-op f16 pipeline);
  * VAE decode alone of realistic latents (the encoder's output), engine and floor;
  * how the decoder carries a latent ERROR: decode(z + delta) - decode(z) in f32 for a delta of the size of the engine's latent error (1.64e-3 rel-L2, white),
    relative to the decoded image -- if this is small, the decoded figure is the decoder's own f16 noise, not accumulated upstream error."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch

torch.set_num_threads(min(16, os.cpu_count() or 8))
import synth
from oracle import sd15 as OS
from tc_light_amd import sd15
from tc_light_amd.vae import VAEEngine

rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
vae = VAEEngine(sd, "cuda")
fr = synth.video_clip(n, 512, 512, seed=12345)["frames"]
t0 = time.time()
with torch.no_grad():
    z_o = OS.vae_encode(sd, fr)
    with OS.half_outputs():
        z_16 = OS.vae_encode(sd, fr)
    z_e = vae.encode_imgs_batch(fr.cuda(), 2)
    print(f"encode alone:  engine {rel(z_e, z_o):.2e}   f16 floor {rel(z_16, z_o):.2e}")
    lat = z_o.half().float()                                    # realistic latents, f16-representable (what the loop hands the decoder)
    x_o = OS.vae_decode(sd, lat)
    with OS.half_outputs():
        x_16 = OS.vae_decode(sd, lat)
    x_e = vae.decode_latents_batch(lat.half().cuda(), 2)
    print(f"decode alone:  engine {rel(x_e, x_o):.2e}   f16 floor {rel(x_16, x_o):.2e}")
    g = torch.Generator().manual_seed(0)
    for eps in (1.64e-3, 5e-3):
        d = torch.randn(lat.shape, generator=g)
        d = d / d.norm() * lat.norm() * eps
        x_p = OS.vae_decode(sd, lat + d)
        print(f"decoder's response to a white latent error of {eps:.2e} rel-L2: decoded rel-L2 {rel(x_p, x_o):.2e}")
print(f"({time.time() - t0:.0f} s)")
