"""Golden vectors for the MemFlowNet sub-networks: the REFERENCE modules (core/Networks/MemFlowNet/cnn.py, sk2.py,
gma.py) loaded with the seeded stand-in weights of tc_light_amd.memflow and run in this container.  Run from the repo root:
python tests/golden/make_golden_memflow_net.py  (needs /root/reference; writes memflow_net.npz; weights are regenerated from seeds by the tests)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, "/root/reference/utils/evaluation/memflow")
from core.Networks.MemFlowNet.cnn import BasicEncoder  # noqa: E402

src = open(os.path.join(ROOT, "tc_light_amd", "memflow.py")).read()
ns = {}
exec(compile(src.replace("from .lib import lib, stream", "lib = stream = None"), "memflow_shapes", "exec"), ns)    # shapes / seeded weights only

out = {}
g = np.random.default_rng(9)
img = torch.from_numpy((g.random((2, 3, 64, 96)) * 2 - 1).astype(np.float32))
out["img"] = img.numpy()
for name, norm, seed in (("fnet", "instance", 21), ("cnet", "batch", 22)):
    m = BasicEncoder(output_dim=256, norm_fn=norm).eval()
    shapes = ns["encoder_param_shapes"]("", norm)
    assert set(shapes) == set(m.state_dict().keys()), (set(shapes) ^ set(m.state_dict().keys()))
    m.load_state_dict(ns["seeded_state_dict"](shapes, seed), strict=True)
    with torch.no_grad():
        out[name] = m(img).numpy()
    out[name + "_seed"] = seed
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "memflow_net.npz"), **out)
print({k: (v.shape, float(np.abs(v).mean())) for k, v in out.items() if hasattr(v, "shape") and v.ndim > 0})

# ---- whole network: the reference MemFlowNet + InferenceCore (inference_core_skflow.py) on seeded weights, three frames
import types  # noqa: E402

for name in ("timm", "timm.models", "timm.models.layers", "timm.models.registry", "timm.models.vision_transformer", "timm.models.helpers"):
    sys.modules.setdefault(name, types.ModuleType(name))      # twins encoder imports (unused by the basicencoder configuration)
sys.path.insert(0, "/root/reference/utils/evaluation")
from memflow.core.Networks.MemFlowNet.MemFlow import MemFlowNet  # noqa: E402
from memflow.core.utils.utils import forward_interpolate  # noqa: E402
from memflow.inference.inference_core_skflow import InferenceCore  # noqa: E402


class Cfg(dict):                                              # stands in for yacs.CfgNode: attribute and item access
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


TAL = (400 * 720 // 64) * 3 / 2                               # configs/things_memflownet.py:30
cfg = Cfg(cnet="basicencoder", fnet="basicencoder", gma="GMA-SK2", pretrain=False, corr_fn="default", corr_levels=4, down_ratio=8, feat_dim=256,
          decoder_depth=12, train_avg_length=TAL)
icfg = Cfg(mem_every=1, enable_long_term=False, enable_long_term_count_usage=True, max_mid_term_frames=2, min_mid_term_frames=1,
           val_decoder_depth=15, train_avg_length=TAL, top_k=None)
SEED = 31
net = MemFlowNet(cfg).eval()
shapes = ns["memflow_param_shapes"]()
assert set(shapes) == set(net.state_dict().keys())
net.load_state_dict(ns["seeded_state_dict"](shapes, SEED), strict=True)
proc = InferenceCore(net, config=icfg)
frames = torch.from_numpy((np.random.default_rng(13).random((4, 3, 128, 192)) * 2 - 1).astype(np.float32))
# smooth the frames a little and make consecutive frames related (shifted copies + noise) so the flow is not pure noise
base = torch.nn.functional.avg_pool2d(frames[0:1], 5, 1, 2)
frames = torch.cat([torch.roll(base, (k, 2 * k), (2, 3)) + 0.05 * frames[k:k + 1] for k in range(4)])
full = {"frames": frames.numpy(), "seed": SEED}
flow_prev = None
with torch.no_grad():
    for i in range(3):
        pair = torch.stack([frames[i], frames[i + 1]])[None]
        low, up = proc.step(pair, end=(i == 2), flow_init=flow_prev)
        full[f"low{i}"], full[f"up{i}"] = low.numpy(), up.numpy()
        flow_prev = forward_interpolate(low[0])[None]          # video_dataparser.py:154 (warm start of the next pair)
        full[f"init{i + 1}"] = flow_prev.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "memflow_full.npz"), **full)
print({k: (v.shape, float(np.abs(v).mean())) for k, v in full.items() if hasattr(v, "shape") and v.ndim > 0})
