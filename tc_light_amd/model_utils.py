"""init_iclight (reference: utils/model_utils.py:12-94) for the MI355X engine.

Loads the SD-1.5 UNet (HF safetensors keys), widens conv_in to 8 input channels with zero-initialised extra weights
(:21-26), ADDS the IC-Light offset file to every tensor (:47-54, strict key match), loads the VAE, and builds the
SDE-DPM-Solver++ scheduler (:71-78).

A missing or mistyped weight path raises FileNotFoundError: the engine never silently relights with noise weights.  Seeded random
tensors of the same architecture (bench / tests: no network, no checkpoints in the images) must be asked for explicitly with
`models.allow_random: true` in the YAML or `TCL_ALLOW_RANDOM_WEIGHTS=1` in the environment.
"""
import os
import warnings

import torch

from . import sd15
from .scheduler import DPMSolverSDEScheduler
from .unet import UNetEngine
from .vae import VAEEngine
from .vidtome import VidToMe


def allow_random(models=None):
    return bool((models or {}).get("allow_random")) or os.environ.get("TCL_ALLOW_RANDOM_WEIGHTS", "0") not in ("", "0")


def _missing(what, path, allow):
    """The one place that decides what a missing checkpoint means."""
    if not allow:
        raise FileNotFoundError(f"{what} weights not found at {path!r}. Point `models.*` of the config at the checkpoint, or opt into "
                                "seeded random stand-in weights with `models.allow_random: true` / TCL_ALLOW_RANDOM_WEIGHTS=1 "
                                "(plumbing and benchmarks only).")
    warnings.warn(f"{what} weights not found at {path!r} -> seeded random stand-ins (allow_random): outputs are not a trained model's")


def _load_safetensors(path):
    from safetensors.torch import load_file
    return load_file(path)


def load_unet_state(unet_path=None, offset_path=None, seed=1, allow=False):
    """utils/model_utils.py:14-54: SD-1.5 UNet state dict with the 8-channel conv_in and the IC-Light offsets added to EVERY key."""
    shapes = sd15.unet_param_shapes()
    if not (unet_path and os.path.exists(unet_path)):
        _missing("SD-1.5 UNet", unet_path, allow)
        return sd15.random_state_dict(shapes, seed)
    sd = {k: v.float() for k, v in _load_safetensors(unet_path).items()}
    w = sd["conv_in.weight"]
    if w.shape[1] == 4:                                    # new_conv_in: zero weights for the 4 concat channels (:21-26)
        w8 = torch.zeros(w.shape[0], 8, w.shape[2], w.shape[3])
        w8[:, :4] = w
        sd["conv_in.weight"] = w8
    if not (offset_path and os.path.exists(offset_path)):
        _missing("IC-Light offset (iclight_sd15_fc.safetensors)", offset_path, allow)
    else:
        off = _load_safetensors(offset_path)
        missing = [k for k in sd if k not in off]
        if missing:
            raise KeyError(f"IC-Light offset file lacks keys {missing[:3]} (reference merges strictly, model_utils.py:50-54)")
        bad = [k for k in sd if tuple(off[k].shape) != tuple(sd[k].shape)]
        if bad:
            raise KeyError(f"IC-Light offset shape mismatch on {bad[:3]}: {tuple(off[bad[0]].shape)} vs {tuple(sd[bad[0]].shape)}")
        sd = {k: sd[k] + off[k].float() for k in sd}
    bad = [k for k, s in shapes.items() if k not in sd or tuple(sd[k].shape) != tuple(s)]
    if bad:
        raise KeyError(f"UNet checkpoint mismatch on {bad[:3]}")
    return sd


def load_vae_state(path=None, seed=2, allow=False):
    shapes = sd15.vae_param_shapes()
    if not (path and os.path.exists(path)):
        _missing("AutoencoderKL", path, allow)
        return sd15.random_state_dict(shapes, seed)
    raw = {k: v.float() for k, v in _load_safetensors(path).items()}
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}     # pre-0.15 diffusers VAE attention names
    sd = {}
    for k, v in raw.items():
        for a, b in ren.items():
            k = k.replace(f"attentions.0.{a}.", f"attentions.0.{b}.")
        sd[k] = v.reshape(shapes[k]) if k in shapes and v.numel() == int(torch.tensor(shapes[k]).prod()) else v
    return sd


def load_memflow_state(path=None, seed=4, allow=False):
    """MemFlowNet_things.pth (eval_utils.py:197-248: torch.load(..., weights_only=True), optional 'module.' prefix)."""
    from . import memflow
    if not (path and os.path.exists(path)):
        _missing("MemFlowNet", path, allow)
        return memflow.seeded_state_dict(memflow.memflow_param_shapes(), seed)
    raw = torch.load(path, map_location="cpu", weights_only=True)
    return {(k[7:] if k.startswith("module.") else k): v for k, v in raw.items()}


def load_rmbg_state(path=None, seed=3, allow=False):
    """briaai/RMBG-1.4 weights (`model.safetensors` / `model.pth` with the reference module's keys, generate.py:149)."""
    from . import rmbg
    if not (path and os.path.exists(path)):
        _missing("BriaRMBG", path, allow)
        return rmbg.random_state_dict(seed)
    raw = _load_safetensors(path) if path.endswith(".safetensors") else torch.load(path, map_location="cpu", weights_only=True)
    return {k: v for k, v in raw.items()}


def init_iclight(device="cuda", models=None, seed=12345):
    """-> (pipe-like namespace with .unet/.vae/.scheduler, scheduler, 'iclight')."""
    from types import SimpleNamespace
    m = models or {}
    ok = allow_random(m)
    unet = UNetEngine(load_unet_state(m.get("unet"), m.get("iclight_offset"), allow=ok), device, VidToMe(device, seed=seed))
    vae = VAEEngine(load_vae_state(m.get("vae"), allow=ok), device)
    scheduler = DPMSolverSDEScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012)
    return SimpleNamespace(unet=unet, vae=vae, scheduler=scheduler), scheduler, "iclight"
