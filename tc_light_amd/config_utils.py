"""Config surface of the reference (utils/VidToMe/config_utils.py:6-74): same CLI flags, YAML keys, base_config chaining,
${a.b} interpolation and save_config -- on PyYAML instead of OmegaConf (not installed in the target image)."""
import argparse
import os
import re
from datetime import datetime

import yaml


class Config(dict):
    """dict with attribute access (the subset of OmegaConf the pipeline uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(base, over):
    out = Config(base)
    for k, v in over.items():
        out[k] = _merge(base[k], v) if isinstance(v, dict) and isinstance(base.get(k), dict) else v
    return out


def _resolve(cfg, root=None):
    root = cfg if root is None else root

    def look(path):
        cur = root
        for p in path.split("."):
            cur = cur[p]
        return _sub(cur) if isinstance(cur, str) else cur

    def _sub(s):
        m = re.fullmatch(r"\$\{([\w.]+)\}", s)
        if m:
            return look(m.group(1))
        return re.sub(r"\$\{([\w.]+)\}", lambda mm: str(look(mm.group(1))), s)
    for k, v in list(cfg.items()):
        if isinstance(v, dict):
            _resolve(v, root)
        elif isinstance(v, str) and "${" in v:
            cfg[k] = _sub(v)
    return cfg


def load_yaml_chain(path, base_override=None):
    cfg = _wrap(yaml.safe_load(open(path)))
    cur, cur_path = cfg, path
    if base_override is not None:
        cur["base_config"] = base_override
    while "base_config" in cur and cur["base_config"] != cur_path:
        base = _wrap(yaml.safe_load(open(cur["base_config"])))
        cfg = _merge(base, cfg)
        cur_path, cur = cur["base_config"], base
    return cfg


def load_config(argv=None, print_config=True):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="configs/tclight_default.yaml", help="Config file path")
    ap.add_argument("--base_config", type=str, default=None, help="Base config file path to override")
    ap.add_argument("--input_path", "-i", type=str, default=None, help="path to video, for a fast usage")
    ap.add_argument("--prompt", "-p", type=str, default=None, help="prompt for video relighting, for a fast usage")
    ap.add_argument("--negative_prompt", "-n", type=str, default=None, help="negative prompt, for a fast usage")
    ap.add_argument("--multi_axis", action="store_true", help="use multi-axis denoising, for a fast usage")
    a = ap.parse_args(argv)
    cfg = load_yaml_chain(a.config, a.base_config)
    if a.input_path is not None and cfg.data.scene_type.lower() == "video":
        cfg.data.rgb_path = a.input_path
    if a.multi_axis:
        cfg.generation.alpha_t = 0.01
    if a.negative_prompt is not None:
        cfg.generation.negative_prompt = a.negative_prompt
    if a.prompt is not None or isinstance(cfg.generation.prompt, str):
        prompt = cfg.generation.prompt if a.prompt is None else a.prompt
        video = os.path.splitext(os.path.basename(cfg.data.rgb_path))[0]
        cfg.work_dir = os.path.join(cfg.work_dir, datetime.now().strftime("%m-%d-%Y"), video)
        os.makedirs(cfg.work_dir, exist_ok=True)
        prev = [int(x[-5:]) for x in os.listdir(cfg.work_dir) if x[-5:].isdigit()]
        cfg.generation.prompt = Config({f"{prompt}-{str(max(prev) + 1 if prev else 0).zfill(5)}": prompt})
    if isinstance(cfg.generation.prompt, str):
        cfg.generation.prompt = Config({"edit": cfg.generation.prompt})
    _resolve(cfg)
    if print_config:
        print("[INFO] loaded config:")
        print(yaml.safe_dump(_plain(cfg), sort_keys=False))
    return cfg


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def save_config(config, path, gene=False, inv=False):
    os.makedirs(path, exist_ok=True)
    c = _plain(config)
    if gene:
        c.pop("inversion", None)
    if inv:
        c.pop("generation", None)
    with open(os.path.join(path, "config.yaml"), "w") as f:
        yaml.safe_dump(c, f, sort_keys=False)
