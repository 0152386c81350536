"""BriaRMBG-1.4 (U^2-Net) matting on the device -- SURVEY 8(f) rank 4; reference briarmbg.py:11-462, used by generate.py:147-167.

The reference runs it once per video in the frames' dtype (f32) to get an alpha matte for background compositing.  Here: f32 NCHW,
BatchNorm (eval) + conv bias folded into a per-channel scale/shift of a direct 3x3 (dilated) convolution kernel that also reads the
decoder's channel concat as two sources and adds the RSU residual (csrc/rmbg.hip); max-pool (ceil_mode), bilinear resize
(align_corners=False) and the final sigmoid are the other three kernels.  Only side output d1 is computed -- the only one
generate.py:159 uses (`rmbg(x * 255)[0][0]`).  State-dict keys are the reference module's, so `briaai/RMBG-1.4` weights load as is.
"""
import numpy as np
import torch

from .lib import lib, stream

# (name, kind, in_ch, mid_ch, out_ch): briarmbg.py:357-380
STAGES = [("stage1", 7, 64, 32, 64), ("stage2", 6, 64, 32, 128), ("stage3", 5, 128, 64, 256), ("stage4", 4, 256, 128, 512),
          ("stage5", "4F", 512, 256, 512), ("stage6", "4F", 512, 256, 512), ("stage5d", "4F", 1024, 256, 512),
          ("stage4d", 4, 1024, 128, 256), ("stage3d", 5, 512, 64, 128), ("stage2d", 6, 256, 32, 64), ("stage1d", 7, 128, 16, 64)]


def rsu_convs(kind, cin, mid, cout):
    """[(name, in, out, dilation)] of one RSU block (briarmbg.py:34-68, 116-142, 183-205, 240-258, 287-301)."""
    if kind == "4F":
        return [("rebnconvin", cin, cout, 1), ("rebnconv1", cout, mid, 1), ("rebnconv2", mid, mid, 2), ("rebnconv3", mid, mid, 4),
                ("rebnconv4", mid, mid, 8), ("rebnconv3d", 2 * mid, mid, 4), ("rebnconv2d", 2 * mid, mid, 2), ("rebnconv1d", 2 * mid, cout, 1)]
    L = kind
    cs = [("rebnconvin", cin, cout, 1), ("rebnconv1", cout, mid, 1)]
    cs += [(f"rebnconv{i}", mid, mid, 1) for i in range(2, L)]
    cs += [(f"rebnconv{L}", mid, mid, 2)]
    cs += [(f"rebnconv{i}d", 2 * mid, mid, 1) for i in range(L - 1, 1, -1)]
    cs += [("rebnconv1d", 2 * mid, cout, 1)]
    return cs


def rmbg_param_shapes():
    """{state-dict key: shape} of BriaRMBG (798 entries incl. the BatchNorm buffers)."""
    sh = {"conv_in.weight": (64, 3, 3, 3), "conv_in.bias": (64,)}
    for name, kind, cin, mid, cout in STAGES:
        for cn, ci, co, _ in rsu_convs(kind, cin, mid, cout):
            p = f"{name}.{cn}."
            sh[p + "conv_s1.weight"] = (co, ci, 3, 3); sh[p + "conv_s1.bias"] = (co,)
            for k in ("weight", "bias", "running_mean", "running_var"):
                sh[p + "bn_s1." + k] = (co,)
            sh[p + "bn_s1.num_batches_tracked"] = ()
    for i, c in enumerate((64, 64, 128, 256, 512, 512), 1):
        sh[f"side{i}.weight"] = (1, c, 3, 3); sh[f"side{i}.bias"] = (1,)
    return sh


def random_state_dict(seed=0):
    """Seeded stand-in weights (no checkpoint in the image): He-scaled convs, BN statistics near identity."""
    g = np.random.default_rng(seed)
    sd = {}
    for k, s in rmbg_param_shapes().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(1, dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy((0.5 + g.random(s)).astype(np.float32))
        elif k.endswith("bn_s1.weight"):
            sd[k] = torch.from_numpy((0.8 + 0.4 * g.random(s)).astype(np.float32))
        elif k.endswith("weight"):
            fan = s[1] * 9
            gain = (2.0 / fan) ** 0.5 / (128.0 if k == "conv_in.weight" else 1.0)      # inputs are in [0, 255]: keep activations O(1)
            sd[k] = torch.from_numpy((g.standard_normal(s) * gain).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((0.1 * g.standard_normal(s)).astype(np.float32))
    return sd


class _Conv:
    def __init__(self, w, scale, shift, dil, dev):
        self.cout, self.cin = w.shape[:2]
        self.w = w.reshape(self.cout, -1).t().contiguous().to(dev)      # [Cin*9, Cout]: tap-major, output channels contiguous
        self.scale, self.shift, self.dil = scale.contiguous().to(dev), shift.contiguous().to(dev), dil


class RMBGEngine:
    def __init__(self, state_dict, device):
        self.dev = torch.device(device)
        self.L = lib()
        sd = {k: v.float() for k, v in state_dict.items() if v.dtype.is_floating_point}
        missing = [k for k in rmbg_param_shapes() if k not in state_dict]
        if missing:
            raise KeyError(f"RMBG state dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        d = self.dev
        one = torch.ones(64)
        self.conv_in = _Conv(sd["conv_in.weight"], one, sd["conv_in.bias"], 1, d)
        self.stages = {}
        for name, kind, cin, mid, cout in STAGES:
            cv = {}
            for cn, ci, co, dil in rsu_convs(kind, cin, mid, cout):
                p = f"{name}.{cn}."
                s = sd[p + "bn_s1.weight"] / torch.sqrt(sd[p + "bn_s1.running_var"] + 1e-5)      # BatchNorm2d eval, eps 1e-5
                t = sd[p + "bn_s1.bias"] + (sd[p + "conv_s1.bias"] - sd[p + "bn_s1.running_mean"]) * s
                cv[cn] = _Conv(sd[p + "conv_s1.weight"], s, t, dil, d)
            self.stages[name] = (kind, cv)
        self.side1 = _Conv(sd["side1.weight"], torch.ones(1), sd["side1.bias"], 1, d)

    # ---- ops
    def conv(self, c, x1, x2=None, stride=1, relu=True, resid=None):
        B, C1, H, W = x1.shape
        C2 = x2.shape[1] if x2 is not None else 0
        assert C1 + C2 == c.cin
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(B, c.cout, Ho, Wo, dtype=torch.float32, device=self.dev)
        self.flops = getattr(self, "flops", 0.0) + 2.0 * 9 * c.cin * c.cout * B * Ho * Wo         # algorithmic conv FLOPs executed (bench.py roofline_rmbg)
        self.L.tcl_conv3x3_direct_f32(x1, C1, x2 if x2 is not None else 0, C2, c.w, c.scale, c.shift, resid if resid is not None else 0, y,
                                      B, H, W, c.cout, c.dil, stride, int(relu), stream())
        return y

    def pool(self, x):
        B, C, H, W = x.shape
        y = torch.empty(B, C, (H + 1) // 2, (W + 1) // 2, dtype=torch.float32, device=self.dev)
        self.L.tcl_maxpool2_ceil_f32(x, y, B * C, H, W, stream())
        return y

    def up(self, x, size, sigmoid=False, scale=1.0, clamp=False):
        B, C, H, W = x.shape
        if (H, W) == tuple(size) and not sigmoid and scale == 1.0 and not clamp:
            return x
        y = torch.empty(B, C, size[0], size[1], dtype=torch.float32, device=self.dev)
        self.L.tcl_resize_bilinear_f32(x, y, B * C, H, W, size[0], size[1], float(scale), int(sigmoid), int(clamp), stream())
        return y

    def rsu(self, name, x1, x2=None):
        """RSU-L / RSU-4F forward (briarmbg.py:70-113, 144-180, 207-237, 260-284, 303-318); input = channel concat [x1 | x2]."""
        kind, cv = self.stages[name]
        hxin = self.conv(cv["rebnconvin"], x1, x2)
        if kind == "4F":
            h1 = self.conv(cv["rebnconv1"], hxin)
            h2 = self.conv(cv["rebnconv2"], h1)
            h3 = self.conv(cv["rebnconv3"], h2)
            h4 = self.conv(cv["rebnconv4"], h3)
            h3d = self.conv(cv["rebnconv3d"], h4, h3)
            h2d = self.conv(cv["rebnconv2d"], h3d, h2)
            return self.conv(cv["rebnconv1d"], h2d, h1, resid=hxin)
        L = kind
        hs = [self.conv(cv["rebnconv1"], hxin)]
        for i in range(2, L):
            hs.append(self.conv(cv[f"rebnconv{i}"], self.pool(hs[-1])))
        top = self.conv(cv[f"rebnconv{L}"], hs[-1])
        d = self.conv(cv[f"rebnconv{L - 1}d"], top, hs[-1])
        for i in range(L - 2, 0, -1):
            d = self.up(d, hs[i - 1].shape[2:])
            d = self.conv(cv[f"rebnconv{i}d"], d, hs[i - 1], resid=hxin if i == 1 else None)
        return d

    @torch.no_grad()
    def forward(self, x):
        """x [B,3,H,W] f32 in [0,255] -> sigmoid(d1) [B,1,H,W]  (= BriaRMBG.forward(x)[0][0], briarmbg.py:390-462)."""
        x = x.float().contiguous()
        hxin = self.conv(self.conv_in, x, stride=2, relu=False)
        hx1 = self.rsu("stage1", hxin)
        hx2 = self.rsu("stage2", self.pool(hx1))
        hx3 = self.rsu("stage3", self.pool(hx2))
        hx4 = self.rsu("stage4", self.pool(hx3))
        hx5 = self.rsu("stage5", self.pool(hx4))
        hx6 = self.rsu("stage6", self.pool(hx5))
        hx5d = self.rsu("stage5d", self.up(hx6, hx5.shape[2:]), hx5)
        hx4d = self.rsu("stage4d", self.up(hx5d, hx4.shape[2:]), hx4)
        hx3d = self.rsu("stage3d", self.up(hx4d, hx3.shape[2:]), hx3)
        hx2d = self.rsu("stage2d", self.up(hx3d, hx2.shape[2:]), hx2)
        hx1d = self.rsu("stage1d", self.up(hx2d, hx1.shape[2:]), hx1)
        self.last_features = hx1d                          # forward(x)[1][0] of the reference (parity checks)
        d1 = self.conv(self.side1, hx1d, relu=False)
        return self.up(d1, x.shape[2:], sigmoid=True)

    @torch.no_grad()
    def estimate_alpha(self, frames, batch_size=2):
        """generate.py:151-163: frames [N,3,H,W] in [0,1] -> alpha [N,1,H,W] in [0,1].  The reference builds `resized_size` as
        (64*round(W*s), 64*round(H*s)) and passes it as `size=` (H_out, W_out): the aspect ratio is transposed -- kept as is."""
        N, _, H, W = frames.shape
        s = (256.0 / float(H * W)) ** 0.5
        size = (int(64 * round(W * s)), int(64 * round(H * s)))
        out = []
        for i in range(0, N, batch_size):
            fr = frames[i:i + batch_size].float().contiguous()
            small = self.up(fr, size, scale=255.0)
            a = self.forward(small)
            out.append(self.up(a, (H, W), clamp=True))
        return torch.cat(out)
