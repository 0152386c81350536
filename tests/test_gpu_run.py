"""GPU: run.py end to end on a tiny synthetic clip with NO flow cache: video -> MemFlowNet flows (written in the reference's cache format)
-> soft masks / track ids -> relighting (1 denoising step) -> stage 1/2 -> output.npy + config.yaml; a second run reuses the cache.
Seeded random weights everywhere: checks plumbing and formats, not image quality."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_run_py_with_flow_estimation(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.join(os.path.dirname(__file__), "..")
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.dirname(__file__))
    import synth
    import run
    d = synth.video_clip(4, 192, 256, seed=2)
    vid = tmp_path / "clip.npy"
    np.save(vid, (d["frames"].permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8))
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(f"""base_config: {os.path.join(root, 'configs', 'tclight_default.yaml')}
work_dir: {tmp_path / 'work'}
data: {{rgb_path: {vid}, height: 192, width: 256}}
generation:
  prompt: {{edit: "warm light"}}
  n_timesteps: 1
  alpha_t: 0.01
  frame_range: [0, 4, 1]
post_opt: {{epochs_exposure: 1, epochs: 1, batch_size: 4}}
models: {{allow_random: true}}
""")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        run.main(["--config", str(cfg)])
        for kind in ("future", "past"):
            dd = tmp_path / f"clip_{kind}_flow_memflow"
            files = sorted(os.listdir(dd))
            assert files == [f"{i:04d}.pt" for i in range(4)]
            assert tuple(torch.load(dd / files[1]).shape) == (1, 2, 192, 256)
        assert torch.load(tmp_path / "clip_future_flow_memflow" / "0003.pt").abs().max().item() == 0      # last future flow is zero
        outs = [os.path.join(r, f) for r, _, fs in os.walk(tmp_path / "work") for f in fs if f == "output.npy"]
        assert len(outs) == 1
        out = np.load(outs[0])
        assert out.shape == (4, 192, 256, 3) and out.dtype == np.uint8
        odir = os.path.dirname(outs[0])
        assert os.path.exists(os.path.join(odir, "config.yaml"))
        assert any(f.startswith("output_gt") for f in os.listdir(odir))                       # generate.py:619-625
        assert sorted(os.listdir(os.path.join(odir, "frames"))) == [f"{i:04d}.png" for i in range(4)]     # save_frame: true (default yaml)
        assert os.path.exists(os.path.join(odir, "loss_exposure.npy")) and os.path.exists(os.path.join(odir, "loss_unique_tensor.npy"))
        bad = tmp_path / "strict.yaml"                                                        # without allow_random missing weights are an error
        bad.write_text(cfg.read_text().replace("allow_random: true", "allow_random: false"))
        with pytest.raises(FileNotFoundError):
            run.main(["--config", str(bad)])
        run.main(["--config", str(cfg)])               # second run: flows come from the cache
    finally:
        os.chdir(cwd)
