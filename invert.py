"""invert.py surface of the reference (invert.py:22-332, DDIM inversion for the non-IC-Light VidToMe models).

The IC-Light path that TC-Light ships never runs inversion (reference run.py:13-22 skips it for sd_version == 'iclight'), so this
engine keeps the entry point for drop-in compatibility and states that plainly instead of silently doing nothing."""
import sys

from tc_light_amd.config_utils import load_config


class Inverter:
    def __init__(self, pipe, scheduler, config):
        self.config = config

    def __call__(self, save_path):
        if self.config.sd_version == "iclight":
            print("[INFO] sd_version 'iclight': no inversion needed, latents start from noise (reference run.py:13-22)")
            return
        raise NotImplementedError("DDIM inversion for sd_version 2.1/2.0/1.5/depth is outside tc_light_amd's scope (SURVEY 2.1 #20)")


if __name__ == "__main__":
    cfg = load_config(sys.argv[1:])
    Inverter(None, None, cfg)(cfg.inversion.save_path)
