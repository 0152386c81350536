"""Oracle (CPU, fp32, plain torch) for the SD-1.5 / IC-Light UNet and the AutoencoderKL VAE.
TEST INFRASTRUCTURE -- see oracle/__init__.py.

PARITY UNPINNED: the reference takes this arithmetic from diffusers==0.32.1 (requirements.txt:1), absent from
/root/reference and from this image; the reference has no tests.  This file restates the published architecture
 over HF state-dict key names.  Reference call sites: generate.py:342-347,
utils/model_utils.py:21-40 (8-channel conv_in + concat_conds), utils/VidToMe/vidtome/patch.py:124-201 (ToMeBlock),
utils/VidToMe/generate_utils.py:140-172 (VAE scaling).
"""
import math

import torch
import torch.nn.functional as F

BLOCK_OUT = (320, 640, 1280, 1280)

# ---- f16 noise floor (round 5).  The reference runs this path in torch.float16 (utils/model_utils.py:12-20: `torch_dtype=torch.float16`): every
# op -- conv, norm, activation, Linear, SDPA, residual add -- accumulates in f32 and ROUNDS ITS OUTPUT TO f16.  `with half_outputs():` makes this
# oracle do exactly that (each op's result passes through f16 and back; weights are expected f16-representable already), so that
# |oracle_f16 - oracle_f32| shows what ANY f16 pipeline, the reference's included, sits at against f32 -- the yardstick for the engine's own
# distance from the f32 oracle (tests/test_gpu_unet.py, __graft_entry__.smoke).  Off by default: the f32 oracle is unchanged, bit for bit.
_HALF = [False]


class half_outputs:
    def __enter__(self):
        self.prev = _HALF[0]
        _HALF[0] = True

    def __exit__(self, *a):
        _HALF[0] = self.prev


def _r(x):
    return x.half().float() if _HALF[0] else x


def _conv(x, w, b=None, **kw):
    return _r(F.conv2d(x, w, b, **kw))


def _lin(x, w, b=None):
    return _r(F.linear(x, w, b))


def _silu(x):
    return _r(F.silu(x))


def timestep_embedding(t, dim=320):
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    e = float(t) * freq
    return torch.cat([torch.cos(e), torch.sin(e)])          # flip_sin_to_cos=True, freq_shift=0


def _gn(x, sd, p, eps, groups=32):
    return _r(F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps))


def resblock(sd, p, x, temb, eps=1e-5):
    h = _conv(_silu(_gn(x, sd, p + "norm1", eps)), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    if temb is not None:
        h = _r(h + _lin(_silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])[None, :, None, None])
    h = _conv(_silu(_gn(h, sd, p + "norm2", eps)), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = _conv(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return _r(x + h)


def _attn(q, k, v, heads):
    b, tq, c = q.shape
    d = c // heads
    q, k, v = (t.reshape(t.shape[0], t.shape[1], heads, d).transpose(1, 2) for t in (q, k, v))
    return _r(F.scaled_dot_product_attention(q, k, v)).transpose(1, 2).reshape(b, tq, c)


def transformer(sd, p, x, text_rep, tome, heads=8):
    b, c, h, w = x.shape
    res = x
    hs = _conv(_gn(x, sd, p + "norm", 1e-6), sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    hs = hs.permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = p + "transformer_blocks.0."
    ln = lambda y, i: _r(F.layer_norm(y, (c,), sd[t + f"norm{i}.weight"], sd[t + f"norm{i}.bias"], 1e-5))
    n1 = ln(hs, 1)
    merged, unmerge = tome(p, n1) if tome is not None else (n1, lambda y: y)
    a = _attn(_lin(merged, sd[t + "attn1.to_q.weight"]), _lin(merged, sd[t + "attn1.to_k.weight"]),
              _lin(merged, sd[t + "attn1.to_v.weight"]), heads)
    a = _lin(a, sd[t + "attn1.to_out.0.weight"], sd[t + "attn1.to_out.0.bias"])
    hs = _r(unmerge(a) + hs)
    n2 = ln(hs, 2)
    a = _attn(_lin(n2, sd[t + "attn2.to_q.weight"]), _lin(text_rep, sd[t + "attn2.to_k.weight"]),
              _lin(text_rep, sd[t + "attn2.to_v.weight"]), heads)
    hs = _r(_lin(a, sd[t + "attn2.to_out.0.weight"], sd[t + "attn2.to_out.0.bias"]) + hs)
    f = _lin(ln(hs, 3), sd[t + "ff.net.0.proj.weight"], sd[t + "ff.net.0.proj.bias"])
    a_, g = f.chunk(2, dim=-1)
    hs = _r(_lin(_r(a_ * _r(F.gelu(g))), sd[t + "ff.net.2.weight"], sd[t + "ff.net.2.bias"]) + hs)
    hs = hs.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return _r(_conv(hs, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"]) + res)


def unet_forward(sd, sample, t, text, tome=None):
    """sample [2F, 8, H, W] (latents | concat_conds), text [2, L, 768] (uncond, cond) -> eps [2F, 4, H, W].
    tome(block_prefix, norm1_out [2F,N,C]) -> (merged [B',T,C], unmerge fn) implements the VidToMe hook."""
    Fr = sample.shape[0] // 2
    text_rep = text.repeat_interleave(Fr, dim=0)                      # generate.py:295
    emb = _r(timestep_embedding(t))
    emb = _lin(_silu(_lin(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
               sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    h = _conv(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [h]
    for i in range(4):
        for j in range(2):
            h = resblock(sd, f"down_blocks.{i}.resnets.{j}.", h, emb)
            if i < 3:
                h = transformer(sd, f"down_blocks.{i}.attentions.{j}.", h, text_rep, tome)
            skips.append(h)
        if i < 3:
            h = _conv(h, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(h)
    h = resblock(sd, "mid_block.resnets.0.", h, emb)
    h = transformer(sd, "mid_block.attentions.0.", h, text_rep, tome)
    h = resblock(sd, "mid_block.resnets.1.", h, emb)
    for i in range(4):
        for j in range(3):
            h = resblock(sd, f"up_blocks.{i}.resnets.{j}.", torch.cat([h, skips.pop()], 1), emb)
            if i > 0:
                h = transformer(sd, f"up_blocks.{i}.attentions.{j}.", h, text_rep, tome)
        if i < 3:
            h = F.interpolate(h, size=skips[-1].shape[-2:], mode="nearest")
            h = _conv(h, sd[f"up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = _silu(_gn(h, sd, "conv_norm_out", 1e-5))
    return _conv(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# ------------------------------------------------------------------------------------------------ VAE
def _vae_attn(sd, p, x):
    b, c, h, w = x.shape
    n = _gn(x, sd, p + "group_norm", 1e-6).reshape(b, c, h * w).transpose(1, 2)
    q, k, v = (_lin(n, sd[p + f"{m}.weight"], sd[p + f"{m}.bias"]) for m in ("to_q", "to_k", "to_v"))
    a = _attn(q, k, v, 1)
    a = _lin(a, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return _r(a.transpose(1, 2).reshape(b, c, h, w) + x)


def _vae_mid(sd, side, h):
    h = resblock(sd, f"{side}.mid_block.resnets.0.", h, None, 1e-6)
    h = _vae_attn(sd, f"{side}.mid_block.attentions.0.", h)
    return resblock(sd, f"{side}.mid_block.resnets.1.", h, None, 1e-6)


def vae_encode(sd, imgs):
    """encode_imgs (generate_utils.py:157-163): imgs [B,3,H,W] in [0,1] -> latent mean * 0.18215 [B,4,H/8,W/8]."""
    h = _conv(_r(2 * imgs - 1), sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(4):
        for j in range(2):
            h = resblock(sd, f"encoder.down_blocks.{i}.resnets.{j}.", h, None, 1e-6)
        if i < 3:
            h = _conv(F.pad(h, (0, 1, 0, 1)), sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                      sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = _vae_mid(sd, "encoder", h)
    h = _conv(_silu(_gn(h, sd, "encoder.conv_norm_out", 1e-6)), sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    h = _conv(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return _r(h[:, :4] * 0.18215)


def vae_decode(sd, latents):
    """decode_latents (generate_utils.py:140-146): -> clamp(decode(latents / 0.18215) / 2 + 0.5, 0, 1)."""
    h = _conv(_r(latents / 0.18215), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = _conv(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _vae_mid(sd, "decoder", h)
    for i in range(4):
        for j in range(3):
            h = resblock(sd, f"decoder.up_blocks.{i}.resnets.{j}.", h, None, 1e-6)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = _conv(_silu(_gn(h, sd, "decoder.conv_norm_out", 1e-6)), sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
    return _r(h / 2 + 0.5).clamp(0, 1)
