"""CPU restatement of the MemFlowNet correlation lookup -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: utils/evaluation/memflow/core/Networks/MemFlowNet/corr.py:74-120 (`CorrBlock`: all-pairs volume
corr[b,i,j] = <f1[b,:,i], f2[b,:,j]> / sqrt(D) (`:110-120`), `avg_pool2d(2, 2)` pyramid over the f2 axes (`:86-88`), per level a
(2r+1)^2 window of bilinear samples around coords / 2^l (`:97-106`, `bilinear_sampler` = grid_sample(align_corners=True, zeros),
core/utils/utils.py:65-79)).  Quirk kept: the window offsets come from `meshgrid(dy, dx)` stacked on the last axis and are added to
(x, y), so the FIRST window index offsets x and the second offsets y; channel = level*(2r+1)^2 + a*(2r+1) + b.

Restated WITHOUT the O((HW)^2) volume (what the reference's unused `alt_cuda_corr` / `OLCorrBlock` `:31-71` do): pooling and
bilinear sampling are linear in f2, so level l samples the avg-pooled f2 pyramid; a window needs the (2r+2)^2 integer neighbours of
the centre only.  (The reference's sampler divides by (size - 1): a pyramid level of height or width 1 yields NaN there; sizes with
all levels >= 2 are the contract.)  Pinned by tests/golden/memflow_corr.npz (outputs of the reference's CorrBlock, tests/golden/make_golden_memflow.py).
"""
import math

import torch
import torch.nn.functional as F


def f2_pyramid(f2, num_levels=4):
    """[B,D,H,W] -> list of avg-pooled maps (floor on odd sizes, like avg_pool2d on the volume's f2 axes)."""
    pyr = [f2]
    for _ in range(num_levels - 1):
        pyr.append(F.avg_pool2d(pyr[-1], 2, stride=2))
    return pyr


def corr_lookup(f1, f2, coords, num_levels=4, radius=4):
    """f1, f2 [B,D,H,W] f32; coords [B,2,H,W] (x, y) in f2 pixels -> [B, L*(2r+1)^2, H, W] f32."""
    B, D, H, W = f1.shape
    r, n = radius, 2 * radius + 1
    out = torch.zeros(B, num_levels * n * n, H, W)
    q = f1.permute(0, 2, 3, 1).reshape(B, H * W, D)                          # [B, P, D]
    for l, g in enumerate(f2_pyramid(f2, num_levels)):
        Hl, Wl = g.shape[-2:]
        cx = (coords[:, 0] / 2 ** l).reshape(B, -1)
        cy = (coords[:, 1] / 2 ** l).reshape(B, -1)
        x0, y0 = torch.floor(cx), torch.floor(cy)
        fx, fy = cx - x0, cy - y0
        # dots with the (n+1)^2 integer neighbours  d[iy][ix] = <f1, f2_l[y0-r+iy, x0-r+ix]>  (0 outside the map)
        ix = x0[:, :, None].long() - r + torch.arange(n + 1)[None, None]      # [B,P,n+1]
        iy = y0[:, :, None].long() - r + torch.arange(n + 1)[None, None]
        okx, oky = (ix >= 0) & (ix < Wl), (iy >= 0) & (iy < Hl)
        gf = g.permute(0, 2, 3, 1)                                            # [B,Hl,Wl,D]
        d = torch.zeros(B, H * W, n + 1, n + 1)
        for b in range(B):
            nb = gf[b][iy[b].clamp(0, Hl - 1)[:, :, None], ix[b].clamp(0, Wl - 1)[:, None, :]]    # [P,n+1,n+1,D]
            d[b] = (nb * q[b][:, None, None, :]).sum(-1) * (oky[b][:, :, None] & okx[b][:, None, :])
        fx_, fy_ = fx[:, :, None, None], fy[:, :, None, None]
        # window entry (a, b): x offset a - r, y offset b - r  ->  d[b..b+1][a..a+1]
        w = ((1 - fx_) * (1 - fy_) * d[:, :, :-1, :-1] + fx_ * (1 - fy_) * d[:, :, :-1, 1:]
             + (1 - fx_) * fy_ * d[:, :, 1:, :-1] + fx_ * fy_ * d[:, :, 1:, 1:])           # [B,P,b,a]
        w = w.permute(0, 1, 3, 2).reshape(B, H, W, n * n) / math.sqrt(D)                    # channel a*n + b
        out[:, l * n * n:(l + 1) * n * n] = w.permute(0, 3, 1, 2)
    return out


# ---------------------------------------------------------------------------------------------------------------- encoders
def _norm(sd, p, x, norm):
    """cnn.py:17-31: BatchNorm2d in eval mode (cnet) or InstanceNorm2d without affine (fnet)."""
    if norm == "batch":
        return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)
    return F.instance_norm(x, eps=1e-5)


def _resblock(sd, p, x, norm, stride):                # cnn.py:46-54
    y = F.relu(_norm(sd, p + "norm1.", F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=1), norm))
    y = F.relu(_norm(sd, p + "norm2.", F.conv2d(y, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), norm))
    if stride != 1:
        x = _norm(sd, p + "norm3.", F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride), norm)
    return F.relu(x + y)


def basic_encoder(sd, p, x, norm):
    """BasicEncoder.forward (cnn.py:191-216), eval mode: x [B,3,H,W] -> [B,output_dim,H/8,W/8]."""
    x = F.relu(_norm(sd, p + "norm1.", F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=2, padding=3), norm))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _resblock(sd, p + f"layer{li}.0.", x, norm, stride)
        x = _resblock(sd, p + f"layer{li}.1.", x, norm, 1)
    return F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
