"""CPU: real-checkpoint readiness of init_iclight's loaders (reference: utils/model_utils.py:12-54).

No SD-1.5 / IC-Light checkpoint exists in the build or test images, so what can be pinned is the LAYOUT a real one has:
  * the HF `unet/diffusion_pytorch_model.safetensors` key -> shape list, enumerated here from the grammar of SURVEY Appendix C with templates of
    its own (not through tc_light_amd/sd15.py), must equal `sd15.unet_param_shapes()` key for key -- and the same for the AutoencoderKL;
  * published known answers: the SD-1.5 UNet has 686 tensors / 859 520 964 parameters (4 input channels), the AutoencoderKL 248 tensors /
    83 653 863 parameters -- numbers anyone can read off the model cards; a wrong channel count anywhere changes them;
  * a FULL-SIZE zero-filled safetensors pair (4-channel UNet + 8-channel IC-Light offset file with exactly the same keys) goes through
    `load_unet_state` strictly: conv_in widened to [320, 8, 3, 3], every key merged, shapes equal to what UNetEngine consumes.
"""
import itertools
import os

import numpy as np
import pytest
import torch

from tc_light_amd import model_utils, sd15


def _expand(t):
    """'a.{0,1}.b.{x,y}' -> all combinations."""
    parts, opts = [], []
    i = 0
    while i < len(t):
        if t[i] == "{":
            j = t.index("}", i)
            opts.append(t[i + 1:j].split(",")); parts.append(None); i = j + 1
        else:
            j = t.find("{", i)
            j = len(t) if j < 0 else j
            parts.append(t[i:j]); i = j
    for combo in itertools.product(*opts):
        it = iter(combo)
        yield "".join(p if p is not None else next(it) for p in parts)


def hf_unet_layout():
    """SURVEY Appendix C, spelled as (template, shape rule)."""
    C = [320, 640, 1280, 1280]
    out = {"conv_in.weight": (320, 4, 3, 3), "conv_in.bias": (320,), "conv_norm_out.weight": (320,), "conv_norm_out.bias": (320,),
           "conv_out.weight": (4, 320, 3, 3), "conv_out.bias": (4,),
           "time_embedding.linear_1.weight": (1280, 320), "time_embedding.linear_1.bias": (1280,),
           "time_embedding.linear_2.weight": (1280, 1280), "time_embedding.linear_2.bias": (1280,)}

    def resnet(p, cin, cout):
        out.update({p + "norm1.weight": (cin,), p + "norm1.bias": (cin,), p + "conv1.weight": (cout, cin, 3, 3), p + "conv1.bias": (cout,),
                    p + "time_emb_proj.weight": (cout, 1280), p + "time_emb_proj.bias": (cout,), p + "norm2.weight": (cout,), p + "norm2.bias": (cout,),
                    p + "conv2.weight": (cout, cout, 3, 3), p + "conv2.bias": (cout,)})
        if cin != cout:
            out.update({p + "conv_shortcut.weight": (cout, cin, 1, 1), p + "conv_shortcut.bias": (cout,)})

    def attention(p, c):
        for k in _expand(p + "{norm,transformer_blocks.0.norm1,transformer_blocks.0.norm2,transformer_blocks.0.norm3}.{weight,bias}"):
            out[k] = (c,)
        for k in _expand(p + "{proj_in,proj_out}.weight"):
            out[k] = (c, c, 1, 1)                                   # SD-1.5: conv projections
        for k in _expand(p + "{proj_in,proj_out}.bias"):
            out[k] = (c,)
        t = p + "transformer_blocks.0."
        for k in _expand(t + "attn{1,2}.to_q.weight"):
            out[k] = (c, c)
        for k in _expand(t + "attn1.{to_k,to_v}.weight"):
            out[k] = (c, c)
        for k in _expand(t + "attn2.{to_k,to_v}.weight"):
            out[k] = (c, 768)                                       # cross_attention_dim
        for k in _expand(t + "attn{1,2}.to_out.0.weight"):
            out[k] = (c, c)
        for k in _expand(t + "attn{1,2}.to_out.0.bias"):
            out[k] = (c,)
        out.update({t + "ff.net.0.proj.weight": (8 * c, c), t + "ff.net.0.proj.bias": (8 * c,), t + "ff.net.2.weight": (c, 4 * c), t + "ff.net.2.bias": (c,)})

    for i in range(4):
        for j in range(2):
            resnet(f"down_blocks.{i}.resnets.{j}.", C[i - 1] if (j == 0 and i > 0) else C[i], C[i])
            if i < 3:
                attention(f"down_blocks.{i}.attentions.{j}.", C[i])
        if i < 3:
            out[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (C[i], C[i], 3, 3); out[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (C[i],)
    resnet("mid_block.resnets.0.", 1280, 1280); resnet("mid_block.resnets.1.", 1280, 1280); attention("mid_block.attentions.0.", 1280)
    # up block i (i = 0 deepest) has 3 resnets; skip connections come from the down path in reverse: channel counts popped from
    skips = [320] + [c for i in range(4) for c in ([C[i], C[i]] + ([C[i]] if i < 3 else []))]      # conv_in, then per down block: 2 resnets (+ downsampler)
    prev = 1280
    for i in range(4):
        c = C[3 - i]
        for j in range(3):
            resnet(f"up_blocks.{i}.resnets.{j}.", prev + skips.pop(), c)
            prev = c
            if i > 0:
                attention(f"up_blocks.{i}.attentions.{j}.", c)
        if i < 3:
            out[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); out[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    assert not skips
    return out


def hf_vae_layout():
    ch = [128, 256, 512, 512]
    out = {"quant_conv.weight": (8, 8, 1, 1), "quant_conv.bias": (8,), "post_quant_conv.weight": (4, 4, 1, 1), "post_quant_conv.bias": (4,),
           "encoder.conv_in.weight": (128, 3, 3, 3), "encoder.conv_in.bias": (128,), "encoder.conv_norm_out.weight": (512,), "encoder.conv_norm_out.bias": (512,),
           "encoder.conv_out.weight": (8, 512, 3, 3), "encoder.conv_out.bias": (8,),
           "decoder.conv_in.weight": (512, 4, 3, 3), "decoder.conv_in.bias": (512,), "decoder.conv_norm_out.weight": (128,), "decoder.conv_norm_out.bias": (128,),
           "decoder.conv_out.weight": (3, 128, 3, 3), "decoder.conv_out.bias": (3,)}

    def resnet(p, cin, cout):
        out.update({p + "norm1.weight": (cin,), p + "norm1.bias": (cin,), p + "conv1.weight": (cout, cin, 3, 3), p + "conv1.bias": (cout,),
                    p + "norm2.weight": (cout,), p + "norm2.bias": (cout,), p + "conv2.weight": (cout, cout, 3, 3), p + "conv2.bias": (cout,)})
        if cin != cout:
            out.update({p + "conv_shortcut.weight": (cout, cin, 1, 1), p + "conv_shortcut.bias": (cout,)})

    for side in ("encoder", "decoder"):
        resnet(f"{side}.mid_block.resnets.0.", 512, 512); resnet(f"{side}.mid_block.resnets.1.", 512, 512)
        for k in _expand(side + ".mid_block.attentions.0.{to_q,to_k,to_v,to_out.0}.weight"):
            out[k] = (512, 512)
        for k in _expand(side + ".mid_block.attentions.0.{group_norm.weight,group_norm.bias,to_q.bias,to_k.bias,to_v.bias,to_out.0.bias}"):
            out[k] = (512,)
    cin = 128
    for i, c in enumerate(ch):
        for j in range(2):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", cin, c); cin = c
        if i < 3:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3); out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    cin = 512
    for i, c in enumerate(reversed(ch)):
        for j in range(3):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", cin, c); cin = c
        if i < 3:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    return out


def _count(d):
    return sum(int(np.prod(s)) for s in d.values())


def test_unet_layout_equals_hf_grammar_and_published_counts():
    hf = hf_unet_layout()
    assert len(hf) == 686 and _count(hf) == 859_520_964            # published: SD-1.5 UNet2DConditionModel
    mine = sd15.unet_param_shapes(in_channels=4)
    assert set(mine) == set(hf), (sorted(set(mine) ^ set(hf))[:5])
    assert all(tuple(mine[k]) == tuple(hf[k]) for k in hf), [k for k in hf if tuple(mine[k]) != tuple(hf[k])][:5]
    eng = sd15.unet_param_shapes()                                  # what the engine consumes: the IC-Light widened conv_in
    assert eng["conv_in.weight"] == (320, 8, 3, 3) and {k: v for k, v in eng.items() if k != "conv_in.weight"} == {k: v for k, v in mine.items() if k != "conv_in.weight"}


def test_vae_layout_equals_hf_grammar_and_published_counts():
    hf = hf_vae_layout()
    assert len(hf) == 248 and _count(hf) == 83_653_863             # published: AutoencoderKL (sd-vae / SD-1.5 vae)
    mine = sd15.vae_param_shapes()
    assert set(mine) == set(hf), (sorted(set(mine) ^ set(hf))[:5])
    assert all(tuple(mine[k]) == tuple(hf[k]) for k in hf)


def test_full_size_checkpoint_pair_loads_strict(tmp_path):
    """A zero-filled full-size pair with the HF layout (UNet: 4 input channels, as published; offset file: the same keys with conv_in
    [320,8,3,3], model_utils.py:50-54) through the loader init_iclight uses; one tensor of each file carries a marker so the merge is visible."""
    from safetensors.torch import save_file
    hf = hf_unet_layout()
    base = {k: torch.zeros(s, dtype=torch.float16) for k, s in hf.items()}
    base["mid_block.resnets.0.conv1.bias"][7] = 1.5
    base["conv_in.weight"][3, 2, 1, 1] = 0.25
    off = {k: torch.zeros((320, 8, 3, 3) if k == "conv_in.weight" else s, dtype=torch.float16) for k, s in hf.items()}
    off["mid_block.resnets.0.conv1.bias"][7] = 0.5
    off["conv_in.weight"][3, 6, 1, 1] = -0.125
    pu, po = str(tmp_path / "unet.safetensors"), str(tmp_path / "iclight_sd15_fc.safetensors")
    save_file(base, pu); save_file(off, po)
    del base, off
    sd = model_utils.load_unet_state(pu, po)
    want = sd15.unet_param_shapes()
    assert set(sd) == set(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    assert sd["conv_in.weight"].shape == (320, 8, 3, 3)
    assert float(sd["mid_block.resnets.0.conv1.bias"][7]) == 2.0                                    # W_sd15 + W_offset
    assert float(sd["conv_in.weight"][3, 2, 1, 1]) == 0.25 and float(sd["conv_in.weight"][3, 6, 1, 1]) == -0.125
    assert float(sd["conv_in.weight"][:, 4:].abs().sum()) == 0.125                                  # the widened half is zero + offset
    # strictness: an offset file that lacks one key, or carries the 4-channel conv_in, is refused
    os.remove(po)
    o2 = {k: torch.zeros((320, 8, 3, 3) if k == "conv_in.weight" else s, dtype=torch.float16) for k, s in hf.items() if k != "conv_out.bias"}
    save_file(o2, po)
    with pytest.raises(KeyError):
        model_utils.load_unet_state(pu, po)
    # the VAE: full-size, incl. the pre-0.15 attention names (query/key/value/proj_attn with [512,512,1,1] weights) some SD-1.5 repos still ship
    hv = hf_vae_layout()
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    old = {}
    for k, s in hv.items():
        kk = k
        if ".attentions.0." in k:
            for a, b in ren.items():
                kk = kk.replace(f"attentions.0.{a}.", f"attentions.0.{b}.")
            if kk.endswith(".weight") and len(s) == 2:
                s = (s[0], s[1], 1, 1)
        old[kk] = torch.zeros(s, dtype=torch.float16)
    pv = str(tmp_path / "vae.safetensors")
    save_file(old, pv)
    sv = model_utils.load_vae_state(pv)
    wv = sd15.vae_param_shapes()
    assert set(sv) == set(wv) and all(tuple(sv[k].shape) == tuple(wv[k]) for k in wv)


def test_unversioned_gemm_table_is_ignored_with_a_warning(tmp_path, monkeypatch):
    """a tile table without the `!tcl-gemm-table gfx950 v3` line (e.g. a round-2 file named by TCL_GEMM_TABLE) must be ignored with a
    warning -- the ctypes binding raises on TCL_EINVAL -- not abort engine start-up."""
    from tc_light_amd import unet
    from tc_light_amd.lib import lib
    p = tmp_path / "old_table.txt"
    p.write_text("# tc_light_amd GEMM tile table\n0 4096 320 320 0 0 0 0 0 0 0 1\n")
    monkeypatch.setenv("TCL_GEMM_TABLE", str(p))
    monkeypatch.setattr(unet.load_gemm_table, "done", False, raising=False)
    L = lib()
    n0 = L.tcl_gemm_tune_size()
    with pytest.warns(UserWarning, match="ignored"):
        unet.load_gemm_table(L)
    assert L.tcl_gemm_tune_size() == n0
    monkeypatch.setattr(unet.load_gemm_table, "done", False, raising=False)
    monkeypatch.delenv("TCL_GEMM_TABLE")
    unet.load_gemm_table(L)                                         # the committed table still loads afterwards
    assert L.tcl_gemm_tune_size() > 900
