"""CPU: host-side rows of SURVEY 8(a)/(f) that need no GPU -- init_iclight's weight merge (A1), the chunked prompt encoding (A4), the
video / frame I/O and config surface ((f)3).  Reference lines are cited at each check."""
import os

import numpy as np
import pytest
import torch


# ------------------------------------------------------------------------------------------------------------------ A1
def _tiny_shapes():
    return {"conv_in.weight": (8, 8, 3, 3), "conv_in.bias": (8,), "time_embedding.linear_1.weight": (16, 8),
            "down_blocks.0.resnets.0.conv1.weight": (8, 8, 3, 3)}


def _write(path, sd):
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in sd.items()}, str(path))


def test_init_iclight_offset_merge_and_conv_in_widening(tmp_path, monkeypatch):
    """utils/model_utils.py:21-26 (conv_in 4 -> 8 input channels, extra weights zero) and :47-54 (W = W_sd15 + W_offset on EVERY key,
    strict) on a synthetic safetensors pair (the real files are 3.4 GB and not in the image)."""
    from tc_light_amd import model_utils as MU, sd15
    shapes = _tiny_shapes()
    monkeypatch.setattr(sd15, "unet_param_shapes", lambda: shapes)
    g = torch.Generator().manual_seed(0)
    base = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}
    base["conv_in.weight"] = base["conv_in.weight"][:, :4].contiguous()        # SD-1.5 ships a 4-channel conv_in
    off = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}          # the IC-Light file has the 8-channel conv_in
    _write(tmp_path / "unet.safetensors", {k: v.half() for k, v in base.items()})
    _write(tmp_path / "off.safetensors", {k: v.half() for k, v in off.items()})
    sd = MU.load_unet_state(str(tmp_path / "unet.safetensors"), str(tmp_path / "off.safetensors"))
    assert set(sd) == set(shapes)
    w = sd["conv_in.weight"]
    assert tuple(w.shape) == (8, 8, 3, 3)
    assert torch.equal(w[:, :4], base["conv_in.weight"].half().float() + off["conv_in.weight"][:, :4].half().float())
    assert torch.equal(w[:, 4:], off["conv_in.weight"][:, 4:].half().float())   # zero-initialised extra channels + offset
    for k in shapes:
        if k != "conv_in.weight":
            assert torch.equal(sd[k], base[k].half().float() + off[k].half().float()), k
    # strict key match (model_utils.py:50-54): an offset file lacking a key is an error, so is a shape mismatch
    bad = dict(off); bad.pop("conv_in.bias")
    _write(tmp_path / "bad.safetensors", bad)
    with pytest.raises(KeyError):
        MU.load_unet_state(str(tmp_path / "unet.safetensors"), str(tmp_path / "bad.safetensors"))
    bad = dict(off); bad["conv_in.bias"] = torch.zeros(9)
    _write(tmp_path / "bad2.safetensors", bad)
    with pytest.raises(KeyError):
        MU.load_unet_state(str(tmp_path / "unet.safetensors"), str(tmp_path / "bad2.safetensors"))
    # a checkpoint of the wrong architecture is refused
    wrong = dict(base); wrong["time_embedding.linear_1.weight"] = torch.zeros(4, 4)
    _write(tmp_path / "wrong.safetensors", wrong)
    off2 = dict(off); off2["time_embedding.linear_1.weight"] = torch.zeros(4, 4)
    _write(tmp_path / "off2.safetensors", off2)
    with pytest.raises(KeyError):
        MU.load_unet_state(str(tmp_path / "wrong.safetensors"), str(tmp_path / "off2.safetensors"))


def test_missing_weights_raise_unless_allowed(tmp_path, monkeypatch):
    """a mistyped path must not silently relight with noise weights."""
    from tc_light_amd import model_utils as MU, sd15
    monkeypatch.delenv("TCL_ALLOW_RANDOM_WEIGHTS", raising=False)
    monkeypatch.setattr(sd15, "unet_param_shapes", _tiny_shapes)
    with pytest.raises(FileNotFoundError):
        MU.load_unet_state(str(tmp_path / "nope.safetensors"), None)
    with pytest.raises(FileNotFoundError):
        MU.load_vae_state(str(tmp_path / "nope.safetensors"))
    with pytest.raises(FileNotFoundError):
        MU.load_rmbg_state(None)
    with pytest.raises(FileNotFoundError):
        MU.load_memflow_state("x.pth")
    base = {k: torch.zeros(*s) for k, s in _tiny_shapes().items()}
    _write(tmp_path / "unet.safetensors", base)
    with pytest.raises(FileNotFoundError):                                          # UNet present, IC-Light offset missing
        MU.load_unet_state(str(tmp_path / "unet.safetensors"), str(tmp_path / "missing_offset.safetensors"))
    assert not MU.allow_random({}) and MU.allow_random({"allow_random": True})
    monkeypatch.setenv("TCL_ALLOW_RANDOM_WEIGHTS", "1")
    assert MU.allow_random({})
    with pytest.warns(UserWarning):
        sd = MU.load_unet_state(None, None, allow=True)
    assert set(sd) == set(_tiny_shapes())
    from tc_light_amd.text import encode_prompt_pair
    with pytest.raises(FileNotFoundError):
        encode_prompt_pair("a", "b", "cpu", enc_dir=str(tmp_path / "no_clip"))


# ------------------------------------------------------------------------------------------------------------------ A4
class _Tok:
    model_max_length, bos_token_id, eos_token_id = 77, 49406, 49407

    def __call__(self, txt, truncation=False, add_special_tokens=False):
        assert truncation is False and add_special_tokens is False          # generate.py:108
        return {"input_ids": [1000 + (hash(w) % 5000) for w in txt.split()]}


class _Enc:
    """last_hidden_state[b, i, :] = token id at (b, i): lets the test read back which ids were fed, chunk by chunk."""

    def __init__(self):
        self.seen = []

    def __call__(self, ids):
        from types import SimpleNamespace
        assert ids.dtype == torch.int64 and ids.shape[1] == 77
        self.seen.append(ids.clone())
        return SimpleNamespace(last_hidden_state=ids[..., None].float().repeat(1, 1, 4))


def test_encode_prompt_chunking_bos_eos_tiling():
    """generate.py:98-135: untruncated tokens -> 75-token chunks, each wrapped [BOS] ... [EOS] and padded with EOS to 77; the side with
    fewer chunks is tiled up to the longer one; chunks are laid along the sequence; result cat([uncond, cond])."""
    from tc_light_amd.text import encode_prompt_inner, encode_prompt_pair, tile_and_concat
    tok, enc = _Tok(), _Enc()
    txt = " ".join(f"w{i}" for i in range(160))                # 160 tokens -> chunks of 75, 75, 10
    ids = tok(txt)["input_ids"]
    out = encode_prompt_inner(txt, tok, enc, "cpu")
    assert out.shape == (3, 77, 4)
    fed = enc.seen[0]
    for c in range(3):
        body = ids[75 * c:75 * (c + 1)]
        assert fed[c, 0].item() == tok.bos_token_id
        assert fed[c, 1:1 + len(body)].tolist() == body
        assert (fed[c, 1 + len(body):] == tok.eos_token_id).all() and fed[c, 1 + len(body):].numel() == 76 - len(body)
    # exactly 75 tokens: one chunk, BOS + 75 + EOS, no padding; 76 tokens: a second chunk with one token
    assert encode_prompt_inner(" ".join(["a"] * 75), tok, _Enc(), "cpu").shape[0] == 1
    assert encode_prompt_inner(" ".join(["a"] * 76), tok, _Enc(), "cpu").shape[0] == 2
    # pair: positive 3 chunks, negative 1 chunk -> negative tiled 3x; [2, 3*77, D] = [uncond, cond]
    neg = "x y z"
    pair = encode_prompt_pair(txt, neg, "cpu", tokenizer=tok, text_encoder=_Enc())
    assert pair.shape == (2, 231, 4) and pair.dtype == torch.float16
    uc1 = encode_prompt_inner(neg, tok, _Enc(), "cpu")
    assert torch.equal(pair[0].float(), torch.cat([uc1[0]] * 3).half().float())
    assert torch.equal(pair[1].float(), out.reshape(231, 4).half().float())
    # 2 vs 3 chunks: ceil(3/2) = 2 repeats cut to 3 -> chunks [0, 1, 0]
    a, b = torch.arange(2.)[:, None, None].repeat(1, 77, 1), torch.arange(3.)[:, None, None].repeat(1, 77, 1) + 10
    c, uc = tile_and_concat(a, b)
    assert c.shape == (1, 231, 1) and c[0, ::77, 0].tolist() == [0, 1, 0] and uc[0, ::77, 0].tolist() == [10, 11, 12]


# ------------------------------------------------------------------------------------------------------------------ (f)3
def test_frame_directory_png_loader_and_writer(tmp_path):
    """utils/VidToMe/utils.py:76-80,108-145 + video_dataparser.py:34-41: a directory of PNG/JPG frames loads through PIL with the
    reference's [-1,1] round trip, resized (short side covers) and centre-cropped; save_video / save_frames write what evaluate.py reads."""
    from PIL import Image
    from tc_light_amd import dataparser as DP
    g = np.random.default_rng(0)
    src = g.integers(0, 256, (3, 40, 60, 3), dtype=np.uint8)
    d = tmp_path / "frames"
    d.mkdir()
    for i, fr in enumerate(src):
        Image.fromarray(fr).save(d / f"{i:04d}.png")
    (d / "notes.txt").write_text("ignored")
    p = DP.VideoDataParser({"rgb_path": str(d), "height": 40, "width": 60}, "cpu")
    assert p.n_frames == 3
    fr = p.load_video()
    assert fr.shape == (3, 3, 40, 60) and fr.min() >= 0 and fr.max() <= 1.0 + 1e-6
    want = torch.from_numpy(src).permute(0, 3, 1, 2).float() / 255.0
    assert (fr - want).abs().max() < 1e-6                        # (x*255/127 - 1 + 1)*127/255 == x up to f32 rounding
    # the reference's own example background loads (generate.py:165: data_parser.load_video(path=background_image_path))
    ref_bg = "/root/reference/examples/background"
    if os.path.isdir(ref_bg):
        bg = p.load_video(path=ref_bg)
        assert bg.shape == (1, 3, 40, 60)
    # resize + centre crop: 40x60 -> working size 32x32: scale = max(32/60, 32/40) = 0.8 -> 32x48 -> crop columns 8..40
    p2 = DP.VideoDataParser({"rgb_path": str(d), "height": 32, "width": 32}, "cpu")
    small = p2.load_video(frame_ids=[0, 2])
    assert small.shape == (2, 3, 32, 32)
    full = torch.nn.functional.interpolate(want[[0, 2]], size=(32, 48), mode="bilinear", antialias=True, align_corners=False)
    assert (small - full[..., 8:40]).abs().max() < 2e-2          # ([-1,1] round trip commutes with the linear resize up to rounding)
    # dark uint8 clips stay uint8 clips (scale from dtype, not from the value range)
    np.save(tmp_path / "dark.npy", np.ones((2, 8, 8, 3), np.uint8))
    assert abs(DP.read_frames(str(tmp_path / "dark.npy")).max().item() - 1 / 255) < 1e-7
    # writer: frames as PNG + output(.mp4 | .npy); ground truth copy with post_fix "_gt" (generate.py:619-625)
    out = tmp_path / "out"
    path = DP.save_video(fr, str(out), save_frame=True, gif=False, fps=25)
    assert os.path.exists(path) and os.path.basename(path).startswith("output")
    assert sorted(os.listdir(out / "frames")) == ["0000.png", "0001.png", "0002.png"]
    back = np.asarray(Image.open(out / "frames" / "0001.png"))
    assert np.abs(back.astype(int) - src[1].astype(int)).max() <= 1
    gt = DP.save_video(fr, str(out), gif=False, post_fix="_gt")
    assert "output_gt" in os.path.basename(gt)
    # whatever container came out (.mp4 with an H.264 encoder, else Motion-JPEG .avi written through PIL) reads back as the same clip
    if path.endswith((".avi", ".mp4")):
        rb = DP.read_frames(path)
        assert rb.shape == fr.shape and (rb - fr.clamp(0, 1)).abs().mean() < 0.03
    # the codec-free container on its own: RIFF AVI, one JPEG per frame, odd frame sizes, index chunk present
    clip = (np.random.default_rng(0).random((4, 21, 35, 3)) * 255).astype(np.uint8)
    DP.write_mjpeg_avi(str(tmp_path / "t.avi"), clip, fps=7, quality=98)
    raw = open(tmp_path / "t.avi", "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and b"MJPG" in raw[:256] and b"idx1" in raw and int.from_bytes(raw[4:8], "little") == len(raw) - 8
    rb = DP.read_mjpeg_avi(str(tmp_path / "t.avi"))
    assert rb.shape == (4, 21, 35, 3) and np.abs(rb.numpy().astype(int) - clip.astype(int)).mean() < 12       # noise is JPEG's worst case
    assert DP.read_mjpeg_avi(str(tmp_path / "dark.npy")) is None
    gif = DP.save_video(fr, str(out), gif=True)
    assert gif.endswith("output.gif") and DP.read_frames(gif).shape == (3, 3, 40, 60)
    DP.save_loss_curve([0.3, 0.2, 0.1], str(out), "loss_exposure")
    assert np.load(out / "loss_exposure.npy").shape == (3,)


def test_frame_ids_and_example_configs():
    """utils/VidToMe/utils.py:330-346 and the reference's six YAMLs (configs/, configs/examples/): same keys load through load_config."""
    from tc_light_amd.config_utils import load_config
    from tc_light_amd.dataparser import get_frame_ids
    assert get_frame_ids([0, -1, 2], 9) == [0, 2, 4, 6, 8]
    assert get_frame_ids([0, 30, 1], 8) == list(range(8))
    assert get_frame_ids([0, 30, 1], 100, frame_ids=[5, 1, 3]) == [1, 3, 5]
    root = os.path.join(os.path.dirname(__file__), "..")
    cwd = os.getcwd()
    os.chdir(root)
    try:
        for y, (h, w, lmr, bg) in {"configs/tclight_default.yaml": (720, 960, 0.6, False), "configs/tclight_custom.yaml": (720, 960, 0.6, False),
                                   "configs/examples/tclight_droid.yaml": (536, 960, 0.6, False),
                                   "configs/examples/tclight_navsim.yaml": (536, 960, 0.6, False),
                                   "configs/examples/tclight_scand.yaml": (536, 960, 0.6, False),
                                   "configs/examples/tclight_bkgd_robotwin.yaml": (480, 640, 0.9, True)}.items():
            c = load_config(["--config", y], print_config=False)
            assert (c.data.height, c.data.width) == (h, w) and c.generation.local_merge_ratio == lmr, y
            assert bool(c.generation.background_cond) == bg and c.post_opt.epochs == 70 and c.sd_version == "iclight"
        c = load_config(["--config", "configs/examples/tclight_bkgd_robotwin.yaml"], print_config=False)
        assert c.generation.global_merge_ratio == 0.8 and c.generation.background_image_path == "examples/background"
        assert c.generation.output_path == "workdir/examples"           # ${work_dir} interpolation through the base config
    finally:
        os.chdir(cwd)


def test_generator_forwards_vidtome_settings():
    """Round 3: chunk_size > target_stride (multi-round local merge, patch.py:43-56) and align_batch=False (merge.py:109-118) are built, so
    the Generator accepts them and forwards align_batch to the patch arguments like apply_patch does (generate_utils.py:98-100)."""
    from types import SimpleNamespace
    from tc_light_amd.generate import Generator
    from tc_light_amd.vidtome import VidToMe
    stub = SimpleNamespace(dev="cpu", tome=SimpleNamespace(args=dict(target_stride=4, align_batch=True)))
    Generator(stub, None, dict(chunk_size=8))
    Generator(stub, None, dict(align_batch=False))
    assert stub.tome.args["align_batch"] is False
    # the randframe rounds of a long chunk: the dst frames of one round are the frames of the next (8 -> 2 -> 1, 16 -> 4 -> 1, 6 -> 1 | 2)
    rf = VidToMe.round_frames(SimpleNamespace(args=dict(target_stride=4), rng=None), 8, [3, 0])
    assert rf == ([8, 2], [3, 0])
    assert VidToMe.round_frames(SimpleNamespace(args=dict(target_stride=4), rng=None), 16, [0, 3])[0] == [16, 4]
    assert VidToMe.round_frames(SimpleNamespace(args=dict(target_stride=4), rng=None), 6, [2])[0] == [6]
    assert VidToMe.round_frames(SimpleNamespace(args=dict(target_stride=4), rng=None), 6, [1, 0])[0] == [6, 2]
    assert VidToMe.round_frames(SimpleNamespace(args=dict(target_stride=4), rng=None), 1)[0] == []
