"""init_iclight (reference: utils/model_utils.py:12-94) for the MI355X engine.

Loads the SD-1.5 UNet (HF safetensors keys), widens conv_in to 8 input channels with zero-initialised extra weights
(:21-26), ADDS the IC-Light offset file to every tensor (:47-54, strict key match), loads the VAE, and builds the
SDE-DPM-Solver++ scheduler (:71-78).  Without the weight files (no network in the build/bench images) seeded random
tensors of the same architecture are used and a warning is printed.
"""
import os
import warnings

import torch

from . import sd15
from .scheduler import DPMSolverSDEScheduler
from .unet import UNetEngine
from .vae import VAEEngine
from .vidtome import VidToMe


def _load_safetensors(path):
    from safetensors.torch import load_file
    return load_file(path)


def load_unet_state(unet_path=None, offset_path=None, seed=1):
    shapes = sd15.unet_param_shapes()
    if unet_path and os.path.exists(unet_path):
        sd = {k: v.float() for k, v in _load_safetensors(unet_path).items()}
        w = sd["conv_in.weight"]
        if w.shape[1] == 4:                                    # new_conv_in: zero weights for the 4 concat channels
            w8 = torch.zeros(w.shape[0], 8, 3, 3)
            w8[:, :4] = w
            sd["conv_in.weight"] = w8
        if offset_path and os.path.exists(offset_path):
            off = _load_safetensors(offset_path)
            missing = [k for k in sd if k not in off]
            if missing:
                raise KeyError(f"IC-Light offset file lacks keys {missing[:3]} (reference merges strictly)")
            sd = {k: sd[k] + off[k].float() for k in sd}
        else:
            warnings.warn(f"IC-Light offset file {offset_path} not found: running the plain SD-1.5 UNet")
        bad = [k for k, s in shapes.items() if k not in sd or tuple(sd[k].shape) != tuple(s)]
        if bad:
            raise KeyError(f"UNet checkpoint mismatch on {bad[:3]}")
        return sd
    warnings.warn("UNet weights not found -> seeded random SD-1.5-shaped weights (outputs are not a trained model's)")
    return sd15.random_state_dict(shapes, seed)


def load_vae_state(path=None, seed=2):
    shapes = sd15.vae_param_shapes()
    if path and os.path.exists(path):
        raw = {k: v.float() for k, v in _load_safetensors(path).items()}
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}     # pre-0.15 diffusers VAE attention names
        sd = {}
        for k, v in raw.items():
            for a, b in ren.items():
                k = k.replace(f"attentions.0.{a}.", f"attentions.0.{b}.")
            sd[k] = v.reshape(shapes[k]) if k in shapes and v.numel() == int(torch.tensor(shapes[k]).prod()) else v
        return sd
    warnings.warn("VAE weights not found -> seeded random AutoencoderKL-shaped weights")
    return sd15.random_state_dict(shapes, seed)


def load_memflow_state(path=None, seed=4):
    """MemFlowNet_things.pth (eval_utils.py:197-248: torch.load(..., weights_only=True), optional 'module.' prefix) or seeded stand-ins."""
    from . import memflow
    if path and os.path.exists(path):
        raw = torch.load(path, map_location="cpu", weights_only=True)
        return {(k[7:] if k.startswith("module.") else k): v for k, v in raw.items()}
    warnings.warn("MemFlowNet weights not found -> seeded random MemFlowNet-shaped weights (the flow is not a trained model's)")
    return memflow.seeded_state_dict(memflow.memflow_param_shapes(), seed)


def load_rmbg_state(path=None, seed=3):
    """briaai/RMBG-1.4 weights (`model.safetensors` / `model.pth` with the reference module's keys, generate.py:149) or seeded stand-ins."""
    from . import rmbg
    if path and os.path.exists(path):
        raw = _load_safetensors(path) if path.endswith(".safetensors") else torch.load(path, map_location="cpu")
        return {k: v for k, v in raw.items()}
    warnings.warn("RMBG weights not found -> seeded random BriaRMBG-shaped weights (the matte is not a trained model's)")
    return rmbg.random_state_dict(seed)


def init_iclight(device="cuda", models=None, seed=12345):
    """-> (pipe-like namespace with .unet/.vae/.scheduler, scheduler, 'iclight')."""
    from types import SimpleNamespace
    m = models or {}
    unet = UNetEngine(load_unet_state(m.get("unet"), m.get("iclight_offset")), device, VidToMe(device, seed=seed))
    vae = VAEEngine(load_vae_state(m.get("vae")), device)
    scheduler = DPMSolverSDEScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012)
    return SimpleNamespace(unet=unet, vae=vae, scheduler=scheduler), scheduler, "iclight"
