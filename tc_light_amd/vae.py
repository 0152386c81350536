"""AutoencoderKL (SD-1.5 VAE) encode / decode on the HIP C ABI.

Mirrors VidToMeGenerator.encode_imgs / decode_latents (utils/VidToMe/generate_utils.py:140-172): imgs [B,3,H,W] f32 in
[0,1] -> latents = mean * 0.18215 [B,4,H/8,W/8] f16, and back to clamp(decode(z / 0.18215) / 2 + 0.5, 0, 1) f32.
NHWC f16 activations; convs through the implicit-GEMM kernel; the single-head 512-d mid-block attention runs as
GEMM (scores) -> row softmax -> GEMM since head_dim 512 exceeds the flash kernel's register budget and it is <1 % of VAE FLOPs.
"""
import torch

from .lib import lib, stream
from . import sd15
from .unet import Ops, _dev, _conv_w

H16 = torch.float16


class VAEEngine:
    scaling = 0.18215       # hard-coded in the reference (generate_utils.py:143,162)

    def __init__(self, state_dict, device):
        self.dev = torch.device(device)
        self.ops = Ops(self.dev)
        self.L = lib()
        sd, d = state_dict, self.dev
        exp = sd15.vae_param_shapes()
        missing = [k for k in exp if k not in sd]
        if missing:
            raise KeyError(f"VAE state dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        self.res = {}
        for k in exp:
            if k.endswith("norm1.weight"):
                p = k[:-len("norm1.weight")]
                r = dict(n1=(_dev(sd[p + "norm1.weight"], d), _dev(sd[p + "norm1.bias"], d)), c1=_conv_w(sd[p + "conv1.weight"], d),
                         b1=_dev(sd[p + "conv1.bias"], d), n2=(_dev(sd[p + "norm2.weight"], d), _dev(sd[p + "norm2.bias"], d)),
                         c2=_conv_w(sd[p + "conv2.weight"], d), b2=_dev(sd[p + "conv2.bias"], d),
                         cin=sd[p + "conv1.weight"].shape[1], cout=sd[p + "conv1.weight"].shape[0])
                if p + "conv_shortcut.weight" in sd:
                    r["sc"] = (_dev(sd[p + "conv_shortcut.weight"].flatten(1), d), _dev(sd[p + "conv_shortcut.bias"], d))
                self.res[p] = r
        self.attn = {}
        for side in ("encoder", "decoder"):
            p = f"{side}.mid_block.attentions.0."
            self.attn[side] = dict(
                gn=(_dev(sd[p + "group_norm.weight"], d), _dev(sd[p + "group_norm.bias"], d)),
                qkv=_dev(torch.cat([sd[p + "to_q.weight"], sd[p + "to_k.weight"], sd[p + "to_v.weight"]]), d),
                bqkv=_dev(torch.cat([sd[p + "to_q.bias"], sd[p + "to_k.bias"], sd[p + "to_v.bias"]]), d),
                out=(_dev(sd[p + "to_out.0.weight"], d), _dev(sd[p + "to_out.0.bias"], d)))
        w = {}

        def small_in(key, cout):       # 3x3 conv with tiny Cin: input carried as 8 channels, K = 72 padded to 128
            cw = sd[key + ".weight"]
            t = torch.zeros(cout, 3, 3, 8)
            t[..., :cw.shape[1]] = cw.permute(0, 2, 3, 1)
            wp = torch.zeros(cout, 128)
            wp[:, :72] = t.reshape(cout, 72)
            return _dev(wp, d), _dev(sd[key + ".bias"], d)
        w["enc_in"] = small_in("encoder.conv_in", 128)
        w["dec_in"] = small_in("decoder.conv_in", 512)
        for i in range(3):
            w[f"down{i}"] = (_conv_w(sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], d), _dev(sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], d))
            w[f"up{i}"] = (_conv_w(sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], d), _dev(sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], d))
        for side, c in (("encoder", 512), ("decoder", 128)):
            w[side + "_no"] = (_dev(sd[side + ".conv_norm_out.weight"], d), _dev(sd[side + ".conv_norm_out.bias"], d))
            w[side + "_out"] = (_conv_w(sd[side + ".conv_out.weight"], d), _dev(sd[side + ".conv_out.bias"], d))
        w["quant"] = (_dev(sd["quant_conv.weight"].flatten(1), d), _dev(sd["quant_conv.bias"], d))
        w["post_quant"] = (_dev(sd["post_quant_conv.weight"].flatten(1), d), _dev(sd["post_quant_conv.bias"], d))
        self.w = w

    # ---------------------------------------------------------------- blocks
    def _res(self, p, x, B, Hh, Ww):
        o, r = self.ops, self.res[p]
        hn = o.groupnorm(x, r["cin"], *r["n1"], B, Hh * Ww, 1e-6, True)
        h1, _, _ = o.conv3x3(hn, B, Hh, Ww, r["cin"], r["c1"], r["b1"])
        hn2 = o.groupnorm(h1, r["cout"], *r["n2"], B, Hh * Ww, 1e-6, True)
        xs = o.gemm(x, r["sc"][0], r["sc"][1]) if "sc" in r else x
        out, _, _ = o.conv3x3(hn2, B, Hh, Ww, r["cout"], r["c2"], r["b2"], resid=xs)
        return out

    def _attn(self, side, x, B, T):
        o, L, a, C = self.ops, self.L, self.attn[side], 512
        n = o.groupnorm(x, C, *a["gn"], B, T, 1e-6, False)
        qkv = o.gemm(n, a["qkv"], a["bqkv"])                                # [B*T, 1536]
        Tp = (T + 63) // 64 * 64
        att = o.empty(B * T, C)
        S = torch.zeros(T, Tp, dtype=H16, device=self.dev)
        vT = torch.zeros(C, Tp, dtype=H16, device=self.dev)
        for b in range(B):
            qb = qkv[b * T:(b + 1) * T]
            o.gemm(qb, qb[:, C:], out=S, M=T, N=T, K=C, lda=3 * C, ldw=3 * C, ldc=Tp)          # S = q k^T
            L.tcl_softmax_rows_f16(S, T, T, Tp, C ** -0.5, stream())
            L.tcl_transpose_f16(qb[:, 2 * C:], vT, 1, T, C, 3 * C, Tp, stream())
            o.gemm(S, vT, out=att[b * T:(b + 1) * T], M=T, N=C, K=Tp, lda=Tp, ldw=Tp, ldc=C)   # P v
        return o.gemm(att, a["out"][0], a["out"][1], resid=x)

    def _mid(self, side, h, B, hh, ww):
        h = self._res(f"{side}.mid_block.resnets.0.", h, B, hh, ww)
        h = self._attn(side, h, B, hh * ww)
        return self._res(f"{side}.mid_block.resnets.1.", h, B, hh, ww)

    # ---------------------------------------------------------------- encode / decode (one batch)
    def encode(self, imgs):
        """imgs [B,3,H,W] f32 device in [0,1] -> [B,4,H/8,W/8] f16 (posterior mean * 0.18215)."""
        o, L, w = self.ops, self.L, self.w
        B, _, Hh, Ww = imgs.shape
        x8 = o.empty(B * Hh * Ww, 8)
        L.tcl_img_to_nhwc8_f16(imgs.contiguous(), x8, B, Hh * Ww, stream())
        col = o.empty(B * Hh * Ww, 128)
        L.tcl_im2col3x3_small_f16(x8, col, B, Hh, Ww, 8, 128, stream())
        h = o.gemm(col, *w["enc_in"])
        del col
        hh, ww = Hh, Ww
        for i in range(4):
            for j in range(2):
                h = self._res(f"encoder.down_blocks.{i}.resnets.{j}.", h, B, hh, ww)
            if i < 3:
                c = (128, 256, 512)[i]
                h, hh, ww = o.conv3x3(h, B, hh, ww, c, *w[f"down{i}"], stride=2, pad=0)
        h = self._mid("encoder", h, B, hh, ww)
        hn = o.groupnorm(h, 512, *w["encoder_no"], B, hh * ww, 1e-6, True)
        m8, _, _ = o.conv3x3(hn, B, hh, ww, 512, *w["encoder_out"])
        q8 = o.empty(B * hh * ww, 8)
        L.tcl_conv1x1_small_f16(m8, 8, *w["quant"], q8, 8, B * hh * ww, 8, 8, stream())
        z = o.empty(B, 4, hh, ww)
        L.tcl_nhwc_to_nchw_f16(q8, 8, z, B, 4, hh * ww, self.scaling, stream())
        return z

    def decode(self, z):
        """z [B,4,h,w] f16 device -> imgs [B,3,8h,8w] f32 in [0,1]."""
        o, L, w = self.ops, self.L, self.w
        B, _, hh, ww = z.shape
        z8 = o.empty(B * hh * ww, 8)
        L.tcl_nchw_to_nhwc_f16(z.contiguous(), z8, 8, B, 4, hh * ww, 1.0 / self.scaling, stream())
        p8 = o.empty(B * hh * ww, 8)
        L.tcl_conv1x1_small_f16(z8, 8, *w["post_quant"], p8, 8, B * hh * ww, 4, 4, stream())
        col = o.empty(B * hh * ww, 128)
        L.tcl_im2col3x3_small_f16(p8, col, B, hh, ww, 8, 128, stream())
        h = o.gemm(col, *w["dec_in"])
        h = self._mid("decoder", h, B, hh, ww)
        for i, c in enumerate((512, 512, 256, 128)):
            for j in range(3):
                h = self._res(f"decoder.up_blocks.{i}.resnets.{j}.", h, B, hh, ww)
            if i < 3:
                h, hh, ww = o.conv3x3(h, B, hh, ww, c, *w[f"up{i}"], up=(2 * hh, 2 * ww))
        hn = o.groupnorm(h, 128, *w["decoder_no"], B, hh * ww, 1e-6, True)
        y, _, _ = o.conv3x3(hn, B, hh, ww, 128, *w["decoder_out"])
        img = torch.empty(B, 3, hh, ww, dtype=torch.float32, device=self.dev)
        L.tcl_nhwc_to_img_f32(y, 3, img, B, hh * ww, stream())
        return img

    # ---------------------------------------------------------------- the reference's batching (batch_size = 2)
    def encode_imgs_batch(self, imgs, batch_size=2):
        return torch.cat([self.encode(b) for b in imgs.split(batch_size, dim=0)])

    def decode_latents_batch(self, latents, batch_size=2):
        return torch.cat([self.decode(b) for b in latents.split(batch_size, dim=0)])
