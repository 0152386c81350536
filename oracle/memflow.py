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


# ---------------------------------------------------------------------------------------------------------------- update block
def pcblock(sd, p, x, k_conv):
    """PCBlock4_Deep_nopool_res.forward (sk2.py:24-30)."""
    def ffn(q, t):
        t = F.gelu(F.conv2d(t, sd[q + "0.weight"], sd[q + "0.bias"]))
        return F.conv2d(t, sd[q + "2.weight"], sd[q + "2.bias"])
    x = F.gelu(x + ffn(p + "ffn1.", x))
    for i, k in enumerate(k_conv):
        x = F.gelu(x + F.conv2d(x, sd[p + f"conv_list.{i}.weight"], sd[p + f"conv_list.{i}.bias"], padding=k // 2, groups=x.shape[1]))
    x = F.gelu(x + F.conv2d(x, sd[p + "pw.weight"], sd[p + "pw.bias"]))
    return ffn(p + "ffn2.", x)


K_CONV, K_GRU = (1, 15), (1, 7)            # sk2.py:201-202


def motion_and_value(sd, p, flow, corr):
    """SKUpdateBlock6_..._Mem_skflow.get_motion_and_value (sk2.py:216-219) with its encoder (sk2.py:112-128); p = 'update_block.'."""
    e = p + "encoder."
    cor = pcblock(sd, e + "convc2.", F.gelu(pcblock(sd, e + "convc1.", corr, K_CONV)), K_CONV)
    flo = pcblock(sd, e + "convf2.", F.conv2d(flow, sd[e + "convf1.weight"], sd[e + "convf1.bias"]), K_CONV)
    out = pcblock(sd, e + "conv.", torch.cat([cor, flo], 1), K_CONV)
    mf = torch.cat([out, flow], 1)
    return mf, F.conv2d(mf, sd[p + "aggregator.to_v.weight"])


def update(sd, p, net, inp, mf, mfg):
    """SKUpdateBlock6_..._Mem_skflow.forward (sk2.py:221-229) -> (net, 0.25*mask, delta_flow)."""
    net = pcblock(sd, p + "gru.", torch.cat([net, inp, mf, mfg], 1), K_GRU)
    delta = pcblock(sd, p + "flow_head.", net, K_CONV)
    m = F.conv2d(F.relu(F.conv2d(net, sd[p + "mask.0.weight"], sd[p + "mask.0.bias"], padding=1)), sd[p + "mask.2.weight"], sd[p + "mask.2.bias"])
    return net, 0.25 * m, delta


def encode_context(sd, img):
    """MemFlowNet.encode_context (MemFlow.py:96-128) for one frame: -> query, key, net, inp  [B,128,h,w] each."""
    c = basic_encoder(sd, "cnet.", img, "batch")
    net, inp = torch.tanh(c[:, :128]), torch.relu(c[:, 128:])
    q, k = F.conv2d(inp, sd["att.to_qk.weight"]).chunk(2, dim=1)
    return q, k, net, inp


def memory_read(query, mem_key, mem_value, scale, train_avg_length):
    """MemoryManager.match_memory without flash-attn (memory_manager_skflow.py:41-88): softmax over the memory axis of
    <query, key> * scale * log(T, train_avg_length); keys/values [B,C,T] (working memory followed by the current frame's)."""
    B, C, h, w = query.shape
    q = query.flatten(2)
    s = scale * math.log(mem_key.shape[-1], train_avg_length)
    sim = torch.einsum("bcl,bct->btl", q, mem_key) * s
    aff = torch.softmax(sim, dim=1)
    return (mem_value @ aff).view(B, -1, h, w)


def upsample_flow(flow, mask):
    """MemFlowNet.upsample_flow (MemFlow.py:172-183)."""
    N, _, H, W = flow.shape
    mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, 8 * H, 8 * W)


class InferenceCore:
    """inference/inference_core_skflow.py:6-54 + MemoryManager (memory_manager_skflow.py) for the things_memflownet configuration
    (mem_every 1, no long-term memory, max / min mid-term frames 2 / 1, decoder depth `iters`)."""

    def __init__(self, sd, iters=15, train_avg_length=(400 * 720 // 64) * 3 / 2, max_mt=2, min_mt=1):
        self.sd, self.iters, self.tal, self.max_mt, self.min_mt = sd, iters, train_avg_length, max_mt, min_mt
        self.mk = self.mv = None

    def step(self, images, end=False, flow_init=None):
        """images [1,2,3,H,W] in [-1,1] -> (flow_low [1,2,H/8,W/8], flow_up [1,2,H,W])."""
        sd = self.sd
        query, key, net, inp = encode_context(sd, images[:, 0])
        fm = basic_encoder(sd, "fnet.", images.flatten(0, 1), "instance").float()
        f1, f2 = fm[0:1], fm[1:2]
        B, _, h, w = f1.shape
        ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        coords0 = torch.stack([xs, ys])[None]
        coords1 = coords0.clone() if flow_init is None else coords0 + flow_init
        scale = 128 ** -0.5
        gamma = sd["update_block.aggregator.gamma"]
        for _ in range(self.iters):
            corr = corr_lookup(f1, f2, coords1)
            mf, val = motion_and_value(sd, "update_block.", coords1 - coords0, corr)
            k_all = key.flatten(2) if self.mk is None else torch.cat([self.mk, key.flatten(2)], -1)
            v_all = val.flatten(2) if self.mv is None else torch.cat([self.mv, val.flatten(2)], -1)
            mfg = mf + gamma * memory_read(query, k_all, v_all, scale, self.tal)
            net, mask, delta = update(sd, "update_block.", net, inp, mf, mfg)
            coords1 = coords1 + delta
        flow_up = upsample_flow(coords1 - coords0, mask)
        if not end:                                            # mem_every = 1: every frame is a memory frame
            self.mk = key.flatten(2) if self.mk is None else torch.cat([self.mk, key.flatten(2)], -1)
            self.mv = val.flatten(2) if self.mv is None else torch.cat([self.mv, val.flatten(2)], -1)
            if self.mk.shape[-1] >= self.max_mt * h * w:       # compress_features: keep the last min_mt frames
                self.mk, self.mv = self.mk[:, :, -self.min_mt * h * w:], self.mv[:, :, -self.min_mt * h * w:]
        return coords1 - coords0, flow_up
