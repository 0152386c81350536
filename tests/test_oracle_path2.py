"""CPU: the oracle (oracle/path2.py) against the golden vectors produced by the reference's own code."""
import numpy as np
import torch
import torch.nn.functional as F

import synth
from oracle import path2 as O


def sub(t, k=97):
    return t.detach().reshape(-1)[::k].numpy()


def test_known_answers(golden):
    g = golden("path2")
    torch.manual_seed(0)
    x = torch.rand(2, 3, 200, 208)
    y = (x + 0.05 * torch.randn_like(x)).clamp(0, 1)
    assert abs(float(O.relaxed_ms_ssim(x, y)) - float(g["ka_msssim"])) < 1e-6
    assert abs(float(O.tv_loss(x, 0.05)) - float(g["ka_tv"])) < 1e-7
    assert abs(float(O.l1_loss(x, y)) - float(g["ka_l1"])) < 1e-7
    assert abs(O.expon_lr(50, 0.01, 0.001, 100) - float(g["ka_lr50"])) < 1e-12
    lr = [O.expon_lr(s, 0.01, 0.001, 70) for s in range(0, 75, 5)]
    np.testing.assert_allclose(lr, g["lr_curve"], rtol=1e-12)


def test_leaves(golden):
    g = golden("path2")
    d = synth.video_clip(4, 176, 192, seed=11)
    ed, fl = d["edited"], d["past_flows"]
    fr = ed.clone().requires_grad_(True)
    wv = O.warp_flow(fr, fl * 3.0)
    gsel = torch.from_numpy(np.random.default_rng(5).standard_normal(wv.shape).astype(np.float32))
    (wv * gsel).sum().backward()
    np.testing.assert_allclose(sub(wv), g["warp_fwd"], atol=1e-6)
    np.testing.assert_allclose(sub(fr.grad), g["warp_grad"], atol=1e-5)
    for tag, (hh, ww) in {"a": (176, 192), "b": (181, 203)}.items():
        xa = ed[:2, :, :hh, :ww] if hh <= 176 else F.interpolate(ed[:2], size=(hh, ww), mode="bilinear")
        xa = xa.clone().requires_grad_(True)
        ya = (xa.detach() * 0.9 + 0.05).clamp(0, 1) + 0.02 * torch.from_numpy(
            np.random.default_rng(6).standard_normal(xa.shape).astype(np.float32))
        v = O.relaxed_ms_ssim(xa, ya)
        v.backward()
        assert abs(float(v) - float(g[f"msssim_{tag}"])) < 1e-6
        np.testing.assert_allclose(sub(xa.grad), g[f"msssim_{tag}_grad"], atol=1e-9, rtol=1e-4)
    xt = ed[:2].clone().requires_grad_(True)
    tv = O.tv_loss(xt, 0.05)
    tv.backward()
    assert abs(float(tv) - float(g["tv"])) < 1e-9
    np.testing.assert_allclose(sub(xt.grad), g["tv_grad"], atol=1e-10, rtol=1e-5)


def test_producer(golden):
    g = golden("path2")
    d5 = synth.video_clip(5, 48, 64, seed=21, shift=(1.0, 0.0), jitter=0.0)
    fwd = -d5["past_flows"].roll(-1, 0)
    fwd[-1] = 0
    sm = O.get_soft_mask_bwds(d5["frames"], fwd, d5["past_flows"], alpha=0.5)
    np.testing.assert_allclose(sm.numpy(), g["softmask"], atol=1e-6)
    ids = O.get_flowid(d5["frames"], fwd, sm, rgb_threshold=0.01)
    assert np.array_equal(ids.numpy(), g["flowid"])           # bit-exact integer work
    assert bool(g["unq_inv_equal_ids"])
    assert torch.equal(O.voxelization_time_only(ids), ids.reshape(-1))


def test_stage1_stage2_three_iterations(golden):
    g = golden("path2")
    d = synth.video_clip(4, 176, 192, seed=11)
    ed, fl, mk = d["edited"], d["past_flows"], d["masks"]
    n, bs = 4, 2
    bts = synth.batches(n, bs, epochs=2, seed=7)[:3]
    bts1 = [torch.tensor(b) for b in ([2, 1], [3, 2], [1, 3])]   # see make_golden.py: frame 0 never 'current'
    # oracle's loop wants len(batches) % epochs == 0 for the lr index; emulate 2 epochs of 2 iters
    img, expo, losses = _stage1(ed, fl, mk, bts1, n, bs)
    np.testing.assert_allclose(losses, g["s1_losses"], rtol=2e-6)
    np.testing.assert_allclose(expo.numpy(), g["s1_exposure"], atol=2e-6)
    np.testing.assert_allclose(sub(img), g["s1_images"], atol=2e-6)
    inv, k = synth.track_ids(n, 176, 192, seed=3)
    _, feats, losses = O.unique_tensor_optimization(ed, inv, fl, mk, bts, bs)
    np.testing.assert_allclose(losses, g["s2_losses"], rtol=2e-6)
    np.testing.assert_allclose(feats.reshape(-1)[::31].numpy(), g["s2_feats"], atol=3e-6)


def _stage1(ed, fl, mk, bts, n, bs):
    exposure = torch.eye(3, 4)[None].repeat(n, 1, 1).requires_grad_(True)
    opt = O.Adam(exposure.data, lr=1e-3)
    losses = []
    for it, idx in enumerate(bts):
        epoch, i = divmod(it, n // bs)
        opt.lr = O.expon_lr(epoch * n // bs + i + 1, 0.01, 0.001, 2 * n // bs)
        loss, _, _ = O.stage1_loss(exposure, ed, idx, fl, mk)
        (gr,) = torch.autograd.grad(loss, exposure)
        losses.append(float(loss))
        opt.step(gr)
    with torch.no_grad():
        out = O.apply_exposure(ed, exposure, torch.arange(n))
    return out, exposure.detach(), losses
