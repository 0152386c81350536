"""CPU restatement of BriaRMBG-1.4 (U^2-Net) inference -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: briarmbg.py (REBNCONV :11-25, RSU7 :34-113, RSU6 :116-180, RSU5 :183-237, RSU4 :240-284, RSU4F :287-318, BriaRMBG :350-462) and
its use in generate.py:147-167.  Functional PyTorch f32 over a state dict with the reference's keys.  Pinned by
tests/golden/rmbg.npz: outputs of the reference module itself loaded with the same seeded state dict (tests/golden/make_golden_rmbg.py).
"""
import torch
import torch.nn.functional as F


def _rebnconv(sd, p, x, dil=1):                      # briarmbg.py:21-25 (BatchNorm in eval mode)
    y = F.conv2d(x, sd[p + "conv_s1.weight"], sd[p + "conv_s1.bias"], padding=dil, dilation=dil)
    y = F.batch_norm(y, sd[p + "bn_s1.running_mean"], sd[p + "bn_s1.running_var"], sd[p + "bn_s1.weight"], sd[p + "bn_s1.bias"], False, 0.0, 1e-5)
    return F.relu(y)


def _up(src, tar):                                   # briarmbg.py:28-31
    return F.interpolate(src, size=tar.shape[2:], mode="bilinear")


def _pool(x):
    return F.max_pool2d(x, 2, stride=2, ceil_mode=True)


def rsu(sd, p, x, L):
    """RSU-L, L in 4..7 (briarmbg.py:70-113 and the shorter variants)."""
    hxin = _rebnconv(sd, p + "rebnconvin.", x)
    hs = [_rebnconv(sd, p + "rebnconv1.", hxin)]
    for i in range(2, L):
        hs.append(_rebnconv(sd, p + f"rebnconv{i}.", _pool(hs[-1])))
    top = _rebnconv(sd, p + f"rebnconv{L}.", hs[-1], dil=2)
    d = _rebnconv(sd, p + f"rebnconv{L - 1}d.", torch.cat((top, hs[-1]), 1))
    for i in range(L - 2, 0, -1):
        d = _rebnconv(sd, p + f"rebnconv{i}d.", torch.cat((_up(d, hs[i - 1]), hs[i - 1]), 1))
    return d + hxin


def rsu4f(sd, p, x):                                 # briarmbg.py:303-318
    hxin = _rebnconv(sd, p + "rebnconvin.", x)
    h1 = _rebnconv(sd, p + "rebnconv1.", hxin)
    h2 = _rebnconv(sd, p + "rebnconv2.", h1, 2)
    h3 = _rebnconv(sd, p + "rebnconv3.", h2, 4)
    h4 = _rebnconv(sd, p + "rebnconv4.", h3, 8)
    h3d = _rebnconv(sd, p + "rebnconv3d.", torch.cat((h4, h3), 1), 4)
    h2d = _rebnconv(sd, p + "rebnconv2d.", torch.cat((h3d, h2), 1), 2)
    return _rebnconv(sd, p + "rebnconv1d.", torch.cat((h2d, h1), 1)) + hxin


def forward_d1(sd, x, with_features=False):
    """x [B,3,H,W] in [0,255] -> sigmoid(d1) [B,1,H,W]  (BriaRMBG.forward(x)[0][0], briarmbg.py:390-462); with_features: also hx1d
    (= forward(x)[1][0], the decoder's last feature map -- an unsaturated quantity for parity checks)."""
    hxin = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], stride=2, padding=1)
    hx1 = rsu(sd, "stage1.", hxin, 7)
    hx2 = rsu(sd, "stage2.", _pool(hx1), 6)
    hx3 = rsu(sd, "stage3.", _pool(hx2), 5)
    hx4 = rsu(sd, "stage4.", _pool(hx3), 4)
    hx5 = rsu4f(sd, "stage5.", _pool(hx4))
    hx6 = rsu4f(sd, "stage6.", _pool(hx5))
    hx5d = rsu4f(sd, "stage5d.", torch.cat((_up(hx6, hx5), hx5), 1))
    hx4d = rsu(sd, "stage4d.", torch.cat((_up(hx5d, hx4), hx4), 1), 4)
    hx3d = rsu(sd, "stage3d.", torch.cat((_up(hx4d, hx3), hx3), 1), 5)
    hx2d = rsu(sd, "stage2d.", torch.cat((_up(hx3d, hx2), hx2), 1), 6)
    hx1d = rsu(sd, "stage1d.", torch.cat((_up(hx2d, hx1), hx1), 1), 7)
    d1 = F.conv2d(hx1d, sd["side1.weight"], sd["side1.bias"], padding=1)
    out = torch.sigmoid(_up(d1, x))
    return (out, hx1d) if with_features else out


def estimate_alpha(sd, frames, batch_size=2):
    """generate.py:151-163 (the transposed `resized_size` included)."""
    N, _, H, W = frames.shape
    s = (256.0 / float(H * W)) ** 0.5
    size = (int(64 * round(W * s)), int(64 * round(H * s)))
    small = F.interpolate(frames, size=size, mode="bilinear")
    a = torch.cat([forward_d1(sd, b * 255.0) for b in small.split(batch_size, dim=0)])
    return F.interpolate(a, size=(H, W), mode="bilinear").clamp(0, 1)
