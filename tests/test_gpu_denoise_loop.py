"""GPU: the whole multi-axis denoising loop (Generator.ddim_sample: xy chunks, yt windows with overlap, CFG, AdaIN, alpha fusion,
SDE-DPM-Solver++ step; reference generate.py:206-284) against the CPU oracle's restatement of the same loop -- oracle UNet
(oracle/sd15.py) as pred_noise inside oracle/pipeline.py's temporal_denoise (pinned to the reference's goldens in
tests/test_oracle_path1.py), oracle scheduler, the same chunk draws and the same SDE noise.
VidToMe is switched off here (the merge decisions are discrete and are covered, with injected indices, by test_gpu_unet.py); what
this test pins is the orchestration around the UNet.  Per step the oracle is fed the latents the engine had at that step and must
reproduce the fused noise prediction handed to the scheduler: rel-L2 <= 1e-2 (f16 engine vs fp32 oracle through ~700 kernels, as in
test_gpu_unet.py; measured 3.3e-3 / 1.5e-3), and the latents after each step must equal the oracle scheduler's update of the same
inputs (rel-L2 <= 2e-3: f16 storage).  (This test found the one-axis up-sampling bug of the implicit conv: a yt plane of <= 4 frames goes 1x1 -> 1x2.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def test_ddim_sample_multi_axis_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import pipeline as OP
    from oracle import sd15 as OS
    from oracle.scheduler import Scheduler as OSch
    from tc_light_amd import hostlogic as HL
    from tc_light_amd import sd15
    from tc_light_amd.generate import Generator
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe

    sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    eng = UNetEngine(sd, "cuda", VidToMe("cuda", seed=5, enabled=False))
    n, hh, ww = 6, 8, 8
    cfg = dict(n_timesteps=2, alpha_t=0.3, final_factor_t=0.5, win_size_t=4, chunk_size=4, guidance_scale=2.0, seed=77, noise_mode="vanilla")
    g = Generator(eng, None, cfg)
    g.prepare_data(torch.zeros(n, 3, 8 * hh, 8 * ww, device="cuda"))
    gen = torch.Generator().manual_seed(3)
    conds = torch.randn(2, 154, 768, generator=gen).half()
    conds_t = torch.randn(2, 77, 768, generator=gen).half()
    cc = torch.randn(n, 4, hh, ww, generator=gen).half()
    x0 = g.init_noise.clone()

    zs = []
    step = g.scheduler.step

    seen = []

    def recording_step(eps, t, x, noise=None, **kw):
        zs.append(None if noise is None else noise.float().cpu())
        seen.append((x.float().cpu(), eps.float().cpu()))          # latents entering the step, fused noise prediction
        return step(eps, t, x, noise=noise, **kw)
    g.scheduler.step = recording_step
    x_hip = g.ddim_sample(x0.clone(), conds.cuda(), conds_t.cuda(), cc.cuda()).float().cpu()
    torch.cuda.synchronize()
    assert torch.isfinite(x_hip).all() and len(zs) == 2 and zs[-1] is None
    # the same run with every chunk in a UNet pass of its own (`max_tokens_per_pass` forces one chunk per group): identical draws
    # and noise, so it may differ only through the row count of the batched kernels (GroupNorm partial sums, K-split rule)
    g1 = Generator(eng, None, dict(cfg, max_tokens_per_pass=1))
    g1.prepare_data(torch.zeros(n, 3, 8 * hh, 8 * ww, device="cuda"))
    x_one = g1.ddim_sample(x0.clone(), conds.cuda(), conds_t.cuda(), cc.cuda()).float().cpu()
    assert rel(x_one, x_hip) < 2e-3

    # ---- the same loop on the CPU oracle
    c = g.cfg
    osch = OSch(c.n_timesteps)
    assert osch.timesteps.tolist() == g.scheduler.timesteps.tolist()
    alphas = OP.alpha_schedule(c.alpha_t, c.final_factor_t, c.n_timesteps)
    xy_s = HL.ChunkSampler(c.seed, c.chunk_size, c.merge_global, c.chunk_ord)          # rank 0: seed + 7919 * 0
    yt_s = HL.ChunkSampler(c.seed + 1, c.chunk_size, c.merge_global, c.chunk_ord)
    x, ccf, text, text_t = x0.float().cpu(), cc.float(), conds.float(), conds_t.float()

    def pred(xin, txt, t):
        return OP.cfg(OS.unet_forward(sd, torch.cat([xin, xin]), t, txt), c.guidance_scale)

    def fused_eps(xc, xy_chunks, yt_chunks, i, t):
        noises = torch.zeros_like(xc)
        for ch in xy_chunks:
            noises[ch] = pred(torch.cat([xc[ch], ccf[ch]], 1), text, float(t))
        return OP.temporal_denoise(xc, ccf, alphas[i], noises, c.win_size_t, [torch.as_tensor(ch) for ch in yt_chunks],
                                   lambda xt, ct, ch, sl: pred(torch.cat([xt, ct], 1), text_t, float(t)))[1]

    nxt = [seen[1][0], x_hip]                                  # latents the engine had after each step
    for i, t in enumerate(osch.timesteps.tolist()):
        xy_chunks, yt_chunks = xy_s.get_chunks(n), yt_s.get_chunks(ww)
        # (a) the engine's own latents at this step -> the fused prediction it handed to the scheduler
        r = rel(seen[i][1], fused_eps(seen[i][0], xy_chunks, yt_chunks, i, t))
        print(f"[denoise loop parity] step {i}: fused eps rel-L2 = {r:.3e}")
        assert r < 1e-2, (i, r)
        # (b) the scheduler update the loop applied to it (multistep state carried by the oracle scheduler)
        z = zs[i] if zs[i] is not None else torch.zeros_like(x)
        r = rel(nxt[i], osch.step(seen[i][1], seen[i][0], z))
        print(f"[denoise loop parity] step {i}: latents after the SDE step rel-L2 = {r:.3e}")
        assert r < 2e-3, (i, r)
