"""CPU: the BriaRMBG oracle (oracle/rmbg.py) against the reference module's own output on the same seeded weights
(tests/golden/rmbg.npz from tests/golden/make_golden_rmbg.py).  f32 both sides: 1e-5 abs on sigmoid outputs."""
import os

import numpy as np
import torch

from oracle import rmbg as OR


def _seeded_sd(seed):
    src = open(os.path.join(os.path.dirname(__file__), "..", "tc_light_amd", "rmbg.py")).read()
    ns = {}
    exec(compile(src.replace("from .lib import lib, stream", "lib = stream = None"), "rmbg_shapes", "exec"), ns)     # no .so needed on CPU
    return ns["random_state_dict"](seed)


def test_rmbg_oracle_vs_reference():
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rmbg.npz"))
    sd = _seeded_sd(int(G["seed"]))
    with torch.no_grad():
        got, feat = OR.forward_d1(sd, torch.from_numpy(G["x"]), with_features=True)
    ref = torch.from_numpy(G["d1"])
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-5
    assert ref.std().item() > 1e-3            # the matte is not constant
    fr = torch.from_numpy(G["hx1d_sub"])
    fs = feat[:, ::4, ::3, ::3]
    assert ((fs - fr).norm() / fr.norm()).item() < 1e-5
