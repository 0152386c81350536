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
