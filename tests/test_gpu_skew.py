"""TCL_SKEW=1: two groups of chunks through the UNet half a transformer block apart (unet.py forward_pair) -- the matching chain of one group
beside the other group's feed-forward / ResNet kernels instead of beside its own attention.  Scheduling only: same bits as the same groups run one after the other."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _two_steps(cap, skew, sd, x, cc, text, chunks, h, w, yt=False):
    from tc_light_amd.generate import Generator
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    dev = x.device
    old = os.environ.get("TCL_SKEW")
    os.environ["TCL_SKEW"] = skew
    try:
        unet = UNetEngine(sd, dev, VidToMe(dev, seed=7))
        gen = Generator(unet, SimpleNamespace(), dict(max_tokens_per_pass=cap, seed=1))
        gen.h, gen.w = h, w
        outs = []
        for t in (801.0, 781.0):                       # two steps: the second one meets the banks the first one left
            noises = torch.zeros_like(x)
            gen._unet_xy(x, cc, chunks, text, t, noises)
            outs.append(noises)
        torch.cuda.synchronize()
        return outs
    finally:
        if old is None:
            del os.environ["TCL_SKEW"]
        else:
            os.environ["TCL_SKEW"] = old


@pytest.mark.parametrize("lens,cap", [((3, 4, 4, 4, 4, 3), 16_000_000),      # one group, cut in two by the skewed runner
                                       ((4, 4, 1, 4, 4, 4, 2), 30_000),       # several capped groups: pairs + a leftover group
                                       ((9, 4, 4), 30_000),                   # a pair and a leftover group; a 9-frame chunk (two local rounds)
                                       ((5,), 16_000_000)])                   # a single chunk: nothing to pair
def test_skewed_pair_equals_plain_pass_bitwise(lens, cap):
    from tc_light_amd import sd15
    dev = torch.device("cuda")
    sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    n, h, w = sum(lens), 32, 48
    g = np.random.default_rng(11)
    x = torch.from_numpy(g.standard_normal((n, 4, h, w)).astype(np.float32)).to(dev).half()
    cc = torch.from_numpy(g.standard_normal((n, 4, h, w)).astype(np.float32)).to(dev).half()
    text = torch.from_numpy(g.standard_normal((2, 154, 768)).astype(np.float32)).to(dev).half()
    chunks, s = [], 0
    for ln in lens:
        chunks.append(list(range(s, s + ln))); s += ln
    chunks = chunks[::-1]                              # any order: the reference permutes them
    # "cut": the same groups of chunks, one pass after the other.  (Against the UNCUT pass the bits only agree at sizes where the GEMMs' K-split rule
    # no longer looks at the row count -- tests/test_gpu_fullsize.py::test_unet_pass_group_size_invariance_beyond_2g_elements covers that.)
    plain = _two_steps(cap, "cut", sd, x, cc, text, chunks, h, w)
    skew = _two_steps(cap, "1", sd, x, cc, text, chunks, h, w)
    for k, (a, b) in enumerate(zip(plain, skew)):
        assert torch.isfinite(a.float()).all()
        d = (a.float() - b.float()).abs().max().item()
        assert torch.equal(a, b), f"step {k}: the skewed pair differs from the plain pass: max |d| = {d:.3e}"
