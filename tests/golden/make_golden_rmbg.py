"""Golden vectors for BriaRMBG: the REFERENCE module (briarmbg.py) loaded with the seeded stand-in state dict of
tc_light_amd.rmbg.random_state_dict(seed) and run in this container.  Run from the repo root: python tests/golden/make_golden_rmbg.py
(needs /root/reference; writes rmbg.npz with the input and sigmoid(d1); the weights are regenerated from the seed by the tests)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from briarmbg import BriaRMBG  # noqa: E402

src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tc_light_amd", "rmbg.py")).read()
ns = {}
exec(compile(src.replace("from .lib import lib, stream", "lib = stream = None"), "rmbg_shapes", "exec"), ns)   # shapes/seeded weights only

SEED = 11
m = BriaRMBG().eval()
m.load_state_dict(ns["random_state_dict"](SEED), strict=True)
g = np.random.default_rng(5)
x = torch.from_numpy((g.random((2, 3, 96, 160)) * 255).astype(np.float32))
with torch.no_grad():
    res = m(x)
    d1, hx1d = res[0][0], res[1][0]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rmbg.npz"), seed=SEED, x=x.numpy(), d1=d1.numpy(),
                    hx1d_sub=hx1d[:, ::4, ::3, ::3].numpy())          # strided subset of the unsaturated last decoder feature map
print(hx1d.shape, float(hx1d.abs().mean()), d1.shape, float(d1.min()), float(d1.max()), float(d1.std()))
