"""Add the GEMM / conv shapes of BASELINE.json configs[3] (60 frames 960x720, VidToMe 0.9 / 0.8, background mode) to a tile table: runs bench.py's
config3_pass (shapes the table lacks are timed on first use) and saves the table.  usage: TCL_GEMM_TABLE=<table> python tools/retune_config3.py <table>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from tc_light_amd import sd15
from tc_light_amd.lib import lib
from tc_light_amd.parallel import Dist
from tc_light_amd.unet import UNetEngine
from tc_light_amd.vae import VAEEngine
from tc_light_amd.vidtome import VidToMe

dev = torch.device("cuda", 0)
unet = UNetEngine(sd15.random_state_dict(sd15.unet_param_shapes(), seed=1), dev, VidToMe(dev, seed=12345))
vae = VAEEngine(sd15.random_state_dict(sd15.vae_param_shapes(), seed=2), dev)
g = np.random.default_rng(5)
conds = torch.from_numpy(g.standard_normal((2, 154, 768)).astype(np.float32)).to(dev).half()
conds_t = torch.from_numpy(g.standard_normal((2, 77, 768)).astype(np.float32)).to(dev).half()
base = dict(alpha_t=0.01, final_factor_t=0.01, batch_size=16, seed=12345)
print(bench.config3_pass(unet, vae, base, conds, conds_t, Dist(), dev))
lib().tcl_gemm_tune_save(sys.argv[1])
