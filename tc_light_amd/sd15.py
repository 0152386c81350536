"""SD-1.5 parameter layout (HF / diffusers key names) for the IC-Light UNet and the AutoencoderKL.

The reference loads `stablediffusionapi/realistic-vision-v51` + the IC-Light offset file with exactly these keys
(utils/model_utils.py:12-54).  Keeping the names means real safetensors load unchanged; without
network access the engine runs on `random_state_dict` (seeded, variance-preserving init) -- architecture and cost are
identical, values are not the trained ones.
"""
import numpy as np
import torch

BLOCK_OUT = (320, 640, 1280, 1280)
CROSS_DIM = 768
HEADS = 8


def _res(sd, p, cin, cout, temb=1280):
    sd[p + "norm1.weight"] = (cin,); sd[p + "norm1.bias"] = (cin,)
    sd[p + "conv1.weight"] = (cout, cin, 3, 3); sd[p + "conv1.bias"] = (cout,)
    if temb:
        sd[p + "time_emb_proj.weight"] = (cout, temb); sd[p + "time_emb_proj.bias"] = (cout,)
    sd[p + "norm2.weight"] = (cout,); sd[p + "norm2.bias"] = (cout,)
    sd[p + "conv2.weight"] = (cout, cout, 3, 3); sd[p + "conv2.bias"] = (cout,)
    if cin != cout:
        sd[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); sd[p + "conv_shortcut.bias"] = (cout,)


def _tfm(sd, p, c):
    sd[p + "norm.weight"] = (c,); sd[p + "norm.bias"] = (c,)
    sd[p + "proj_in.weight"] = (c, c, 1, 1); sd[p + "proj_in.bias"] = (c,)
    t = p + "transformer_blocks.0."
    for n in ("norm1", "norm2", "norm3"):
        sd[t + n + ".weight"] = (c,); sd[t + n + ".bias"] = (c,)
    for a, kd in (("attn1", c), ("attn2", CROSS_DIM)):
        sd[t + a + ".to_q.weight"] = (c, c)
        sd[t + a + ".to_k.weight"] = (c, kd); sd[t + a + ".to_v.weight"] = (c, kd)
        sd[t + a + ".to_out.0.weight"] = (c, c); sd[t + a + ".to_out.0.bias"] = (c,)
    sd[t + "ff.net.0.proj.weight"] = (8 * c, c); sd[t + "ff.net.0.proj.bias"] = (8 * c,)
    sd[t + "ff.net.2.weight"] = (c, 4 * c); sd[t + "ff.net.2.bias"] = (c,)
    sd[p + "proj_out.weight"] = (c, c, 1, 1); sd[p + "proj_out.bias"] = (c,)


def unet_param_shapes(in_channels=8):
    sd = {}
    sd["conv_in.weight"] = (320, in_channels, 3, 3); sd["conv_in.bias"] = (320,)
    sd["time_embedding.linear_1.weight"] = (1280, 320); sd["time_embedding.linear_1.bias"] = (1280,)
    sd["time_embedding.linear_2.weight"] = (1280, 1280); sd["time_embedding.linear_2.bias"] = (1280,)
    cin = 320
    for i, c in enumerate(BLOCK_OUT):
        for j in range(2):
            _res(sd, f"down_blocks.{i}.resnets.{j}.", cin, c)
            cin = c
            if i < 3:
                _tfm(sd, f"down_blocks.{i}.attentions.{j}.", c)
        if i < 3:
            sd[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3); sd[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    _res(sd, "mid_block.resnets.0.", 1280, 1280)
    _tfm(sd, "mid_block.attentions.0.", 1280)
    _res(sd, "mid_block.resnets.1.", 1280, 1280)
    rev = BLOCK_OUT[::-1]
    prev = rev[0]
    for i, c in enumerate(rev):
        skip_in = rev[min(i + 1, 3)]
        for j in range(3):
            sc = skip_in if j == 2 else c
            _res(sd, f"up_blocks.{i}.resnets.{j}.", (prev if j == 0 else c) + sc, c)
            if i > 0:
                _tfm(sd, f"up_blocks.{i}.attentions.{j}.", c)
        prev = c
        if i < 3:
            sd[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); sd[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    sd["conv_norm_out.weight"] = (320,); sd["conv_norm_out.bias"] = (320,)
    sd["conv_out.weight"] = (4, 320, 3, 3); sd["conv_out.bias"] = (4,)
    return sd


def _vae_attn(sd, p, c):
    sd[p + "group_norm.weight"] = (c,); sd[p + "group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sd[p + n + ".weight"] = (c, c); sd[p + n + ".bias"] = (c,)


def vae_param_shapes():
    sd = {}
    ch = (128, 256, 512, 512)
    sd["encoder.conv_in.weight"] = (128, 3, 3, 3); sd["encoder.conv_in.bias"] = (128,)
    cin = 128
    for i, c in enumerate(ch):
        for j in range(2):
            _res(sd, f"encoder.down_blocks.{i}.resnets.{j}.", cin, c, temb=0)
            cin = c
        if i < 3:
            sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3); sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    for side, cc in (("encoder", 512), ("decoder", 512)):
        _res(sd, f"{side}.mid_block.resnets.0.", cc, cc, temb=0)
        _vae_attn(sd, f"{side}.mid_block.attentions.0.", cc)
        _res(sd, f"{side}.mid_block.resnets.1.", cc, cc, temb=0)
    sd["encoder.conv_norm_out.weight"] = (512,); sd["encoder.conv_norm_out.bias"] = (512,)
    sd["encoder.conv_out.weight"] = (8, 512, 3, 3); sd["encoder.conv_out.bias"] = (8,)
    sd["quant_conv.weight"] = (8, 8, 1, 1); sd["quant_conv.bias"] = (8,)
    sd["post_quant_conv.weight"] = (4, 4, 1, 1); sd["post_quant_conv.bias"] = (4,)
    sd["decoder.conv_in.weight"] = (512, 4, 3, 3); sd["decoder.conv_in.bias"] = (512,)
    cin = 512
    for i, c in enumerate(ch[::-1]):
        for j in range(3):
            _res(sd, f"decoder.up_blocks.{i}.resnets.{j}.", cin, c, temb=0)
            cin = c
        if i < 3:
            sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    sd["decoder.conv_norm_out.weight"] = (128,); sd["decoder.conv_norm_out.bias"] = (128,)
    sd["decoder.conv_out.weight"] = (3, 128, 3, 3); sd["decoder.conv_out.bias"] = (3,)
    return sd


def random_state_dict(shapes, seed, gain=1.0):
    """Seeded synthetic weights (numpy PCG64 -> identical in the build container and on the GPU box), f32 rounded to f16
    values so the f32 oracle and the f16 engine see the same numbers.  Matrices ~ N(0, gain/fan_in), norm scales ~ 1."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, s in shapes.items():
        if k.endswith(".weight") and len(s) == 1:
            v = 1.0 + 0.1 * rng.standard_normal(s)
        elif k.endswith(".bias"):
            v = 0.05 * rng.standard_normal(s)
        else:
            fan_in = int(np.prod(s[1:]))
            v = rng.standard_normal(s) * (gain / fan_in) ** 0.5
        out[k] = torch.from_numpy(v.astype(np.float32)).half().float()
    return out
