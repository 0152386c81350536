"""smoke(): one tiny UNet denoise call on cuda:0 checked against the CPU oracle (called from __graft_entry__.smoke)."""
import numpy as np
import torch


def run():
    from oracle import sd15 as OS            # test infrastructure: only used here as the checker
    from . import sd15
    from .unet import UNetEngine
    from .vidtome import VidToMe
    sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    tome = VidToMe("cuda:0", seed=3, enabled=False)       # no merging -> no discrete choices in the comparison
    eng = UNetEngine(sd, "cuda:0", tome)
    g = np.random.default_rng(0)
    F, h, w = 1, 16, 16
    x = torch.from_numpy(g.standard_normal((F, 8, h, w)).astype(np.float32)).half().float()
    text = torch.from_numpy(g.standard_normal((2, 77, 768)).astype(np.float32)).half().float()
    xin = torch.cat([x, x]).permute(0, 2, 3, 1).contiguous().cuda().half()
    eps = eng.forward_nhwc(xin, F, h, w, 801.0, text.cuda().half())
    torch.cuda.synchronize()
    out = eps.view(2 * F, h, w, 4).permute(0, 3, 1, 2).float().cpu()
    with torch.no_grad():
        ref = OS.unet_forward(sd, torch.cat([x, x]), 801.0, text, None)
    r = ((out - ref).norm() / ref.norm()).item()
    assert r < 1e-2, f"UNet parity vs oracle: rel-L2 {r}"
    print(f"[smoke] UNet forward rel-L2 vs oracle = {r:.2e}")
