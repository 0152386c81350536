"""Oracle (CPU, fp32, autograd) for path 2: the two-stage temporal-consistency optimiser.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  All citations are /root/reference paths.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

SH_C0 = 0.28209479177387814  # utils/sh_utils.py:26


def rgb2sh(rgb):  # utils/sh_utils.py:114-115
    return (rgb - 0.5) / SH_C0


def sh2rgb(sh):  # utils/sh_utils.py:116-117
    return sh * SH_C0 + 0.5


def l1_loss(a, b):  # utils/loss_utils.py:25-26
    return (a - b).abs().mean()


def expon_lr(step, lr_init, lr_final, max_steps, delay_steps=0, delay_mult=1.0):
    """utils/general_utils.py:31-64 (log-linear interpolation, optional sine delay)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if delay_steps > 0:
        rate = delay_mult + (1 - delay_mult) * math.sin(0.5 * math.pi * min(max(step / delay_steps, 0.0), 1.0))
    else:
        rate = 1.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def warp_flow(frames, past_flows):
    """utils/flow_utils.py:5-16 -- backward bicubic warp, zeros padding, align_corners."""
    n, _, h, w = frames.shape
    gx = past_flows[:, 0] + torch.arange(w, dtype=past_flows.dtype)
    gy = past_flows[:, 1] + torch.arange(h, dtype=past_flows.dtype)[:, None]
    gx = (gx / (w - 1) - 0.5) * 2
    gy = (gy / (h - 1) - 0.5) * 2
    grid = torch.stack([gx, gy], dim=-1)
    return F.grid_sample(frames, grid, mode="bicubic", padding_mode="zeros", align_corners=True)


def gauss_window(size=11, sigma=1.5):
    """pytorch_msssim `_fspecial_gauss_1d` (third-party, unpinned): normalised 1-D Gaussian."""
    c = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_filter(x, win1d):
    """pytorch_msssim `gaussian_filter`: separable depthwise 'valid' conv, dim skipped if < win."""
    c = x.shape[1]
    k = win1d.numel()
    out = x
    if x.shape[2] >= k:
        out = F.conv2d(out, win1d.view(1, 1, k, 1).repeat(c, 1, 1, 1), groups=c)
    if x.shape[3] >= k:
        out = F.conv2d(out, win1d.view(1, 1, 1, k).repeat(c, 1, 1, 1), groups=c)
    return out


def _ssim_maps(x, y, win, data_range=1.0, k1=0.01, k2=0.03):
    """utils/loss_utils.py:73-123 -> per-(batch,channel) ssim and cs means."""
    c1 = (k1 * data_range) ** 2
    c2 = (k2 * data_range) ** 2
    mu1, mu2 = gaussian_filter(x, win), gaussian_filter(y, win)
    s11 = gaussian_filter(x * x, win) - mu1 * mu1
    s22 = gaussian_filter(y * y, win) - mu2 * mu2
    s12 = gaussian_filter(x * y, win) - mu1 * mu2
    cs = (2 * s12 + c2) / (s11 + s22 + c2)
    ssim = (2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1) * cs
    return ssim.flatten(2).mean(-1), cs.flatten(2).mean(-1)


MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)  # utils/loss_utils.py:181


def relaxed_ms_ssim(x, y, data_range=1.0, start_level=1):
    """utils/loss_utils.py:125-211: levels below start_level contribute ones; relu on cs;
    avg_pool2d(k=2, padding=size%2) between levels; prod of level**weight; mean."""
    assert x.shape == y.shape and x.dim() == 4
    assert min(x.shape[-2:]) > 160, "loss_utils.py:176-179"
    win = gauss_window().to(x)
    w = x.new_tensor(MS_WEIGHTS)
    vals = []
    levels = len(MS_WEIGHTS)
    for i in range(levels):
        if i >= start_level:
            ssim_pc, cs = _ssim_maps(x, y, win, data_range)
        else:
            ssim_pc = torch.ones_like(x[:, :, 0, 0])
            cs = torch.ones_like(x[:, :, 0, 0])
        if i < levels - 1:
            vals.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    vals.append(torch.relu(ssim_pc))
    stack = torch.stack(vals, dim=0)
    return torch.prod(stack ** w.view(-1, 1, 1), dim=0).mean()


def tv_loss(x, weight):
    """utils/loss_utils.py:324-340."""
    b, c, h, w = x.shape
    count_h = c * (h - 1) * w
    count_w = c * h * (w - 1)
    h_tv = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum()
    return weight * 2 * (h_tv / count_h + w_tv / count_w) / b


def scatter_mean(src, index, k=None):
    """torch_scatter.scatter(src, index, dim=0, reduce='mean') (third-party, unpinned):
    rows = max(index)+1, empty rows 0 (count clamped to 1)."""
    if k is None:
        k = int(index.max()) + 1
    out = torch.zeros(k, src.shape[1], dtype=src.dtype)
    out.index_add_(0, index.long(), src)
    cnt = torch.bincount(index.long(), minlength=k).clamp(min=1).to(src.dtype)
    return out / cnt[:, None]


def apply_exposure(images, exposure, idx):
    """generate.py:405-407 / utils/dataloader.py:38-42: per-frame 3x4 affine then clamp."""
    b, _, h, w = images.shape
    flat = images.permute(0, 2, 3, 1).reshape(b, h * w, 3)
    t = torch.bmm(flat, exposure[idx, :3, :3]) + exposure[idx, None, :3, 3]
    return t.clamp(0, 1).reshape(b, h, w, 3).permute(0, 3, 1, 2)


class Adam:
    """torch.optim.Adam semantics (generate.py:381,483-487), single tensor, no weight decay."""

    def __init__(self, p, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps = p, lr, betas[0], betas[1], eps
        self.m = torch.zeros_like(p)
        self.v = torch.zeros_like(p)
        self.t = 0

    def step(self, g):
        self.t += 1
        self.m.mul_(self.b1).add_(g, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        denom = (self.v.sqrt() / math.sqrt(bc2)).add_(self.eps)
        self.p.addcdiv_(self.m, denom, value=-self.lr / bc1)


def stage1_loss(exposure, edited, idx, past_flows, masks, lambda_dssim=0.2, lambda_flow=0.8):
    """One loss evaluation of generate.py:399-430.  edited: full [N,3,H,W]; idx: batch ids."""
    pre_idx = (idx - 1).clamp(min=0)
    cat_idx = torch.cat([idx, pre_idx])
    cat = apply_exposure(edited[cat_idx], exposure, cat_idx)
    images, pre_images = cat[: len(idx)], cat[len(idx):]
    tgt = edited[idx]
    l_photo = l1_loss(images, tgt) * (1 - lambda_dssim) + (1.0 - relaxed_ms_ssim(images, tgt, 1.0, 1)) * lambda_dssim
    warped = warp_flow(pre_images, past_flows[idx])
    valid = idx > 0
    m = masks[idx][valid]
    l_flow = l1_loss(warped[valid] * m, images[valid] * m)
    return (1 - lambda_flow) * l_photo + lambda_flow * l_flow, l_photo, l_flow


def exposure_align(edited, past_flows, masks, batches, epochs, batch_size,
                   lr_init=0.01, lr_final=0.001, lambda_dssim=0.2, lambda_flow=0.8):
    """generate.py:354-451 with the DataLoader shuffle replaced by the explicit `batches`
    (list over iterations of int64 index tensors).  Returns (aligned images, exposure, losses)."""
    n = edited.shape[0]
    exposure = torch.eye(3, 4)[None].repeat(n, 1, 1).requires_grad_(True)
    opt = Adam(exposure.data, lr=1e-3)
    total_iters = epochs * n // batch_size
    per_epoch = len(batches) // epochs
    losses = []
    for it, idx in enumerate(batches):
        epoch, i = divmod(it, per_epoch)
        opt.lr = expon_lr(epoch * n // batch_size + i + 1, lr_init, lr_final, total_iters)
        loss, _, _ = stage1_loss(exposure, edited, idx, past_flows, masks, lambda_dssim, lambda_flow)
        (g,) = torch.autograd.grad(loss, exposure)
        losses.append(float(loss))
        opt.step(g)
    with torch.no_grad():
        out = apply_exposure(edited, exposure, torch.arange(n))
    return out, exposure.detach(), losses


def stage2_loss(features_dc, unq_inv_nhw, idx, target, past_flows, masks,
                lambda_dssim=0.2, lambda_flow=0.8, lambda_tv=0.05):
    """One loss evaluation of generate.py:496-512.  unq_inv_nhw: [N,H,W] int64."""
    n, h, w = unq_inv_nhw.shape
    cat_idx = torch.cat([idx, (idx - 1).clamp(min=0)])
    inv = unq_inv_nhw[cat_idx].reshape(-1)
    cat = torch.index_select(sh2rgb(features_dc), 0, inv).clamp(0, 1)
    cat = cat.reshape(len(cat_idx), h, w, 3).permute(0, 3, 1, 2)
    images, pre_images = cat[: len(idx)], cat[len(idx):]
    warped = warp_flow(pre_images, past_flows[idx])
    valid = idx > 0
    m = masks[idx][valid]
    l_flow = l1_loss(warped[valid] * m, images[valid] * m)
    l_photo = (1.0 - relaxed_ms_ssim(images, target[idx], 1.0, 1)) * lambda_dssim
    return (1 - lambda_flow) * l_photo + lambda_flow * l_flow + tv_loss(images, lambda_tv), l_photo, l_flow


def unique_tensor_optimization(edited, unq_inv, past_flows, masks, batches, batch_size,
                               feature_lr=0.05, lambda_dssim=0.2, lambda_flow=0.8, lambda_tv=0.05):
    """generate.py:453-533 with explicit `batches`.  Returns (images, features_dc, losses)."""
    n, _, h, w = edited.shape
    lr = feature_lr * batch_size / n
    pix = edited.permute(0, 2, 3, 1).reshape(n * h * w, 3)
    feats = rgb2sh(scatter_mean(pix, unq_inv)).contiguous().requires_grad_(True)
    opt = Adam(feats.data, lr=lr, eps=1e-15)
    inv_nhw = unq_inv.reshape(n, h, w).long()
    losses = []
    for idx in batches:
        loss, _, _ = stage2_loss(feats, inv_nhw, idx, edited, past_flows, masks, lambda_dssim, lambda_flow, lambda_tv)
        (g,) = torch.autograd.grad(loss, feats)
        losses.append(float(loss))
        opt.step(g)
    with torch.no_grad():
        img = sh2rgb(feats)[unq_inv.long()].clamp(0, 1).reshape(n, h, w, 3).permute(0, 3, 1, 2)
    return img, feats.detach(), losses


# ---- stage-2 input producer -------------------------------------------

def get_soft_mask_bwds(org_images, flows, past_flows, alpha=0.1, beta=1e2, diff_threshold=0.1):
    """utils/flow_utils.py:40-54 (batching is only a memory device; result is identical)."""
    mask = torch.ones_like(org_images[:, 0])
    pf = past_flows[1:]
    f2b = warp_flow(flows[:-1], pf)
    mask[1:] *= torch.sigmoid(-beta * (torch.linalg.norm(pf + f2b, dim=1)
                                       - (torch.linalg.norm(pf, dim=1) + torch.linalg.norm(f2b, dim=1) + 1) * alpha))
    d = (warp_flow(org_images[:-1], pf) - org_images[1:]).abs().max(dim=1).values
    mask[1:] *= torch.sigmoid(-beta * (d - org_images.max().item() * diff_threshold))
    return mask[:, None]


def get_flowid(frames, flows, mask_bwds, rgb_threshold=0.01):
    """utils/flow_utils.py:56-93.  Sequential over frames; where several source pixels land on
    one target the reference's advanced-index assignment keeps the LAST writer in row-major
    source order on CPU (nondeterministic on GPU) -- this oracle fixes 'last in row-major order'."""
    n, _, h, w = frames.shape
    ids = torch.full((n, h, w), -1, dtype=torch.int64)
    ids[0] = torch.arange(h * w).view(h, w)
    last = h * w
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    thr = frames.max().item() * rgb_threshold
    for i in range(1, n):
        x = (gx + flows[i - 1, 0]).round().long()
        y = (gy + flows[i - 1, 1]).round().long()
        m = (x >= 0) & (x < w) & (y >= 0) & (y < h)
        m &= mask_bwds[i, 0] > 0.5          # NB: tested at *source* coordinates (flow_utils.py:83)
        xs, ys, sx, sy = x[m], y[m], gx[m], gy[m]
        ok = (frames[i, :, ys, xs] - frames[i - 1, :, sy, sx]).abs().max(dim=0).values < thr
        ids[i, ys[ok], xs[ok]] = ids[i - 1, sy[ok], sx[ok]]
        un = ids[i] == -1
        cnt = int(un.sum())
        ids[i][un] = last + torch.arange(cnt)
        last += cnt
    return ids


def voxelization_time_only(flow_ids):
    """utils/general_utils.py:222-256 with voxel_size=None: torch.unique(dim=0) inverse."""
    _, inv = torch.unique(flow_ids.reshape(-1, 1), return_inverse=True, dim=0)
    return inv
