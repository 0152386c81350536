"""CPU: the oracle's f16-output mode (oracle/sd15.py `half_outputs`, round 5) -- the yardstick the GPU parity tests read the engine's distance from the
f32 oracle against.  The reference runs this path in torch.float16 (utils/model_utils.py:12-20): every op rounds its output to f16.  Checked here on the
VAE (small enough for the CPU suite): the mode changes the result by an f16-sized amount, leaves no state behind, and the f32 oracle is bit-unchanged."""
import numpy as np
import torch


def test_half_outputs_is_an_f16_sized_perturbation_and_restores_state():
    from oracle import sd15 as OS
    from tc_light_amd import sd15
    sd = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
    g = np.random.default_rng(0)
    z = torch.from_numpy(g.standard_normal((1, 4, 8, 8)).astype(np.float32)).half().float() * 0.18215
    img = torch.from_numpy(g.random((1, 3, 32, 32)).astype(np.float32))
    with torch.no_grad():
        d32, e32 = OS.vae_decode(sd, z), OS.vae_encode(sd, img)
        with OS.half_outputs():
            assert OS._HALF[0]
            d16, e16 = OS.vae_decode(sd, z), OS.vae_encode(sd, img)
            with OS.half_outputs():          # nests
                pass
            assert OS._HALF[0]
        assert not OS._HALF[0]
        d32b, e32b = OS.vae_decode(sd, z), OS.vae_encode(sd, img)
    assert torch.equal(d32, d32b) and torch.equal(e32, e32b)                 # the f32 oracle is untouched
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    rd, re = rel(d16, d32), rel(e16, e32)
    print(f"f16 noise floor of the VAE oracle: decode {rd:.2e}, encode {re:.2e}")
    assert 5e-5 < rd < 5e-3 and 5e-5 < re < 5e-3                              # ~a few f16 ulps accumulated over ~60 ops: not zero, not large
    # every value the f16 mode hands on is f16-representable
    assert torch.equal(e16, e16.half().float())
