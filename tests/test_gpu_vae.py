"""GPU parity of the VAE engine vs the fp32 CPU oracle (random seeded weights).  Tolerance rel-L2 <= 1e-2 (f16 vs f32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()


def test_vae_encode_decode():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import sd15 as OS
    from tc_light_amd import sd15
    from tc_light_amd.vae import VAEEngine
    sd = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
    vae = VAEEngine(sd, "cuda")
    g = np.random.default_rng(0)
    imgs = torch.from_numpy(g.random((3, 3, 72, 104), dtype=np.float32))
    z = vae.encode_imgs_batch(imgs.cuda())
    zo = OS.vae_encode(sd, imgs)
    r = rel(z, zo)
    print(f"[vae encode] rel-L2 = {r:.3e}")
    assert z.shape == (3, 4, 9, 13) and r < 1e-2
    lat = torch.from_numpy(g.standard_normal((3, 4, 9, 13)).astype(np.float32) * 0.18215).half()
    img = vae.decode_latents_batch(lat.cuda())
    io = OS.vae_decode(sd, lat.float())
    r = rel(img, io)
    print(f"[vae decode] rel-L2 = {r:.3e}")
    assert img.shape == (3, 3, 72, 104) and r < 1e-2
    assert float(img.min()) >= 0 and float(img.max()) <= 1
