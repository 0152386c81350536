"""GPU: END-TO-END parity of the whole path -- Generator.__call__ (reference generate.py:560-611: VAE encode -> denoising loop with VidToMe
ON through the default block-major `forward_many` schedule -> VAE decode -> stage 1 -> stage 2) against the same composition on the CPU
oracle (tests/e2e_oracle.py) with identical seeds: same synthetic frames, seeded weights, initial noise, chunk draws, VidToMe draws, SDE
noise and mini-batch schedules.

test_config1_end_to_end  = BASELINE.json configs[0] IN FULL: 8 frames 512x512, 4 denoising steps, single axis (alpha_t = 0), VidToMe 0.6/0.5,
                           stage 1 35 epochs + stage 2 70 epochs (TCL_E2E_EPOCHS="a,b" shortens the optimiser for quick local runs).
test_multi_axis_bank_carry_over = a small multi-axis run with VidToMe ON: pins that the global-token banks the xy pass leaves behind are
                           the ones the yt pass of the same step starts from (reset only in post_iter, generate_utils.py:235-238).

Tolerances (north_star: 1e-3 rel-L2 on the output).  The engine computes in f16 with f32 accumulation, the oracle in f32; the UNet/VAE
arithmetic of the oracle is parity-unpinned w.r.t. diffusers (oracle/sd15.py header).  With the engine's merge maps injected into the
oracle the relit frames out of denoise + decode agree to 1e-3 rel-L2 (measured 9.85e-4, asserted < 1.0e-3 = north_star's figure, and against the
f16 noise floor of the path: the oracle itself with f16 op outputs, measured in the test).  Stage 1/2 then run 105 Adam
iterations whose update is +-lr regardless of the gradient's size: the oracle run twice with inputs differing by 1e-7 ends 1.3e-2 apart
(measured in the test), so after stage 2 the assertion is "within 1.5x the oracle's own self-distance" plus agreement of every
iteration's loss.  With the oracle's own matching the discrete decisions differ on near-tied f16 scores: figure printed, bounded loosely.
"""
import os
import time

import numpy as np
import pytest
import torch

import synth
import e2e_oracle as E

pytestmark = pytest.mark.gpu


def _engines(vidtome_seed=12345):
    from tc_light_amd import sd15
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vae import VAEEngine
    from tc_light_amd.vidtome import VidToMe
    sd_unet = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    sd_vae = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
    return sd_unet, sd_vae, UNetEngine(sd_unet, "cuda", VidToMe("cuda", seed=vidtome_seed)), VAEEngine(sd_vae, "cuda")


def _text(seed, L):
    g = np.random.default_rng(seed)
    return torch.from_numpy(g.standard_normal((2, L, 768)).astype(np.float32)).half()


def test_config1_end_to_end():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd.generate import Generator
    e1, e2 = (int(v) for v in os.environ.get("TCL_E2E_EPOCHS", "35,70").split(","))
    n, H, W = 8, 512, 512
    cfg = dict(n_timesteps=4, alpha_t=0.0, epochs_exposure=e1, epochs=e2, batch_size=16, seed=12345)
    sd_unet, sd_vae, unet, vae = _engines()
    d = synth.video_clip(n, H, W, seed=12345)
    inv, k = synth.track_ids(n, H, W, seed=3)
    conds, conds_t = _text(5, 154), _text(6, 77)
    gen = Generator(unet, vae, cfg)
    rec = E.Recorder(gen)
    stages = {}
    enc, dec = vae.encode_imgs_batch, vae.decode_latents_batch
    vae.encode_imgs_batch = lambda x, bs: stages.setdefault("cc", enc(x, bs))
    vae.decode_latents_batch = lambda z, bs: stages.setdefault("clean", dec(stages.setdefault("lat", z.clone()), bs))
    t0 = time.time()
    out, info = gen(d["frames"].cuda(), conds.cuda(), conds_t.cuda(), d["past_flows"].cuda(), d["masks"].cuda(), inv.cuda().int(), n_total=n, k=k)
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    rec.finish()
    vae.encode_imgs_batch, vae.decode_latents_batch = enc, dec
    assert torch.isfinite(out).all() and len(rec.zs) == 4
    n_merge_events = len(rec.traces)
    assert n_merge_events == 10 * len(rec.draws) and len(rec.draws) >= 4 * 2           # 10 merging blocks x chunks x steps

    # ------------------------------------------------------------------ the oracle, same seeds
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    t0 = time.time()
    c = gen.cfg
    cc = E.vae_batches(E.OS.vae_encode, sd_vae, d["frames"])
    x0 = gen.init_noise.float().cpu()
    tome_i = E.InjectedToMe(rec.traces)
    with torch.no_grad():
        lat_i = E.oracle_denoise(sd_unet, x0, cc, conds.float(), conds_t.float(), c, tome_i, rec.zs, c.seed, c.seed + 1)
    assert tome_i.exhausted()
    clean_i = E.vae_batches(E.OS.vae_decode, sd_vae, lat_i)
    t_den = time.time() - t0
    # ... and the f16 NOISE FLOOR of this path (round 5): the same oracle composition with every op's output rounded to f16 -- what the reference's
    # own torch.float16 pipeline does (oracle/sd15.py `half_outputs`) -- against the f32 oracle, same maps, same draws.  The engine's distance
    # from the f32 oracle is to be read against THIS figure: it fuses norm + activation and keeps f32 accumulators across fused ops, so it
    # rounds at fewer points than an op-by-op f16 pipeline.
    with E.OS.half_outputs(), torch.no_grad():
        cc16 = E.vae_batches(E.OS.vae_encode, sd_vae, d["frames"])
        tome_16 = E.InjectedToMe(rec.traces)
        lat16 = E.oracle_denoise(sd_unet, x0, cc16, conds.float(), conds_t.float(), c, tome_16, rec.zs, c.seed, c.seed + 1)
        clean16 = E.vae_batches(E.OS.vae_decode, sd_vae, lat16)
    floor = dict(encode=E.rel(cc16, cc), latents=E.rel(lat16, lat_i), decoded=E.rel(clean16, clean_i))
    print("[e2e config 1] f16 noise floor (oracle with f16 op outputs vs f32 oracle): " + ", ".join(f"{k_} {v:.2e}" for k_, v in floor.items()))
    _, final_i, l1, l2 = E.oracle_post_opt(clean_i, d["past_flows"], d["masks"], inv, c, n)
    # the same two optimiser stages on the oracle, started from the ENGINE's decoded frames: separates the engine's stage-1/2 arithmetic from
    # the conditioning of the reference's algorithm (Adam's first steps are +-lr whatever the gradient's size: 0.05*16/8 * C0 = 0.028 RGB per
    # step here, so a 1e-3 input difference is amplified -- the oracle fed with 1e-3-perturbed inputs moves by 1.2-1.4e-2 rel-L2 itself)
    _, final_h, l1s, l2s = E.oracle_post_opt(stages["clean"].cpu(), d["past_flows"], d["masks"], inv, c, n)
    # ... and the oracle against ITSELF with its input perturbed at the f32 rounding level (1e-7 relative): the reference's optimiser is
    # chaotic at the pixel level -- Adam's update is +-lr whatever the gradient's size, so rounding noise decides the sign wherever the
    # gradient nearly cancels and 105 iterations spread that -- which bounds what ANY two implementations can agree to after stage 2
    g7 = torch.Generator().manual_seed(1)
    _, final_p, _, _ = E.oracle_post_opt((clean_i * (1 + 1e-7 * torch.randn(clean_i.shape, generator=g7))).clamp(0, 1), d["past_flows"], d["masks"], inv, c, n)
    t_all = time.time() - t0
    r = dict(encode=E.rel(stages["cc"].cpu(), cc), latents=E.rel(stages["lat"].cpu(), lat_i), decoded=E.rel(stages["clean"].cpu(), clean_i),
             final_same_decoded=E.rel(out.cpu(), final_h), final=E.rel(out.cpu(), final_i), oracle_vs_oracle_1e7=E.rel(final_p, final_i))
    print(f"[e2e config 1, injected maps] HIP {t_hip:.1f} s (cold) vs oracle {t_all:.0f} s on {torch.get_num_threads()} threads (denoise+VAE {t_den:.0f} s); "
          f"rel-L2: " + ", ".join(f"{k_} {v:.2e}" for k_, v in r.items()))
    l1h, l2h = info["losses_exposure"].cpu().numpy(), info["losses_unique"].cpu().numpy()
    print(f"[e2e config 1] stage-1 loss first/last HIP {l1h[0]:.5f}/{l1h[-1]:.5f} oracle {l1[0]:.5f}/{l1[-1]:.5f}; "
          f"stage-2 HIP {l2h[0]:.5f}/{l2h[-1]:.5f} oracle {l2[0]:.5f}/{l2[-1]:.5f}")
    dd = (out.cpu() - final_h).abs()
    print(f"[e2e config 1] final vs oracle-from-same-decoded: median |diff| {dd.median().item():.2e}, fraction > 1e-2: {(dd > 1e-2).float().mean().item():.4f}")
    checks = [r["encode"] < 2e-3, r["latents"] < 5e-3,
              r["decoded"] < 1.0e-3,                   # north_star's 1e-3 rel-L2 on the relit frames out of the denoise + decode path (measured 9.85e-4)
              r["decoded"] < 1.25 * floor["decoded"], r["latents"] < 1.25 * floor["latents"], r["encode"] < 1.25 * floor["encode"],      # ... and no further from f32 than an op-by-op f16 pipeline is
              # after the two optimiser stages no pointwise 1e-3 exists for anyone: the engine must sit within the oracle's own self-distance
              # (x1.5), from the same decoded frames and over the whole path; the per-iteration LOSSES must agree (checked below, 2e-2 per
              # iteration -- measured 1e-5 at the last one)
              r["final_same_decoded"] < 1.5 * max(r["oracle_vs_oracle_1e7"], 2e-3),
              r["final"] < 1.5 * max(r["oracle_vs_oracle_1e7"], 2e-3)]
    # per-iteration LOSSES (105 of them): against the oracle started from the engine's own decoded frames they must agree to rounding (measured
    # 1e-5 relative; asserted 2e-4) -- the chaos above lives in single pixels, global means do not see it; against the oracle's whole path
    # (decoded frames 1e-3 apart) to 2e-3
    loss_ok = (np.allclose(l1h, np.asarray(l1s), rtol=2e-4) and np.allclose(l2h, np.asarray(l2s), rtol=2e-4)
               and np.allclose(l1h, np.asarray(l1), rtol=2e-3) and np.allclose(l2h, np.asarray(l2), rtol=2e-3))
    print(f"[e2e config 1] loss trajectories, max relative difference: same decoded frames stage 1 {np.abs(l1h / np.asarray(l1s) - 1).max():.2e} "
          f"stage 2 {np.abs(l2h / np.asarray(l2s) - 1).max():.2e}; whole path {np.abs(l1h / np.asarray(l1) - 1).max():.2e} / {np.abs(l2h / np.asarray(l2) - 1).max():.2e}")
    # statistics of the FINAL frames that are not chaotic (sign noise of single codebook rows averages out): per-frame mean colour,
    # mean colour of 64 groups of tracks (track id mod 64), and the masked warp error of the result (the flow term of the loss, evaluated
    # on the outputs).  Scale: the oracle's own distance between its two runs 1e-7 apart.
    fl, mk = d["past_flows"], d["masks"]

    def stats(img):
        fm = img.mean(dim=(2, 3))                                                   # [n, 3]
        flat = img.permute(0, 2, 3, 1).reshape(-1, 3)
        grp = (inv.long() % 64)
        gm = torch.zeros(64, 3).index_add_(0, grp, flat) / torch.bincount(grp, minlength=64).clamp_min(1)[:, None]
        warped = E.O2.warp_flow(img[:-1], fl[1:])
        we = ((warped - img[1:]).abs() * mk[1:]).mean()
        return fm, gm, we
    sh, so, sp, si = stats(out.cpu()), stats(final_h), stats(final_p), stats(final_i)
    d_fm, d_gm, d_we = (sh[0] - so[0]).abs().max().item(), (sh[1] - so[1]).abs().max().item(), abs(sh[2] / so[2] - 1).item()
    s_fm, s_gm, s_we = (sp[0] - si[0]).abs().max().item(), (sp[1] - si[1]).abs().max().item(), abs(sp[2] / si[2] - 1).item()
    print(f"[e2e config 1] non-chaotic statistics, engine vs oracle from the same decoded frames (oracle vs itself 1e-7 apart): per-frame mean colour "
          f"{d_fm:.2e} ({s_fm:.2e}), track-group mean colour {d_gm:.2e} ({s_gm:.2e}), masked warp error rel {d_we:.2e} ({s_we:.2e})")
    checks += [d_fm < max(3 * s_fm, 3e-4), d_gm < max(3 * s_gm, 3e-4), d_we < max(3 * s_we, 3e-3)]

    # ------------------------------------------------------------------ the oracle deciding its own matches
    # (the first 2 of the 4 steps by default -- 60 merges -- to keep the test inside ~5 minutes of oracle time; TCL_E2E_COMPUTED=4 runs all)
    nc = int(os.environ.get("TCL_E2E_COMPUTED", "2"))
    if nc > 0:
        tome_c = E.ComputedToMe(rec.draws, traces=rec.traces)
        with torch.no_grad():
            lat_c = E.oracle_denoise(sd_unet, x0, cc, conds.float(), conds_t.float(), c, tome_c, rec.zs, c.seed, c.seed + 1, max_steps=nc)
        lat_h = stages["lat"].cpu() if nc >= 4 else rec.seen[nc][0]             # the engine's latents after nc steps
        rc = dict(latents=E.rel(lat_h, lat_c))
        if nc >= 4:
            rc["decoded"] = E.rel(stages["clean"].cpu(), E.vae_batches(E.OS.vae_decode, sd_vae, lat_c))
        ag, ags = np.asarray(tome_c.agree), np.asarray(tome_c.agree_src)
        print(f"[e2e config 1, computed maps, {nc} steps] positions restored from the SAME source token: mean {ags.mean():.3f} min {ags.min():.3f} over {len(ags)} "
              f"merges (raw equality of the stored unmerge maps, whose slot numbering of unmerged tokens differs by design: mean {ag.mean():.3f}); rel-L2: "
              + ", ".join(f"{k_} {v:.2e}" for k_, v in rc.items()))
        # measured (round 5): 0.979 mean / 0.932 min over the 60 merges of 2 steps, on random-weight (near-isotropic) activations -- the raw figure
        # of rounds 2-4 (0.59) mostly counted the differently NUMBERED unmerged slots, not different decisions
        checks.append(rc["latents"] < 2e-2 and ags.mean() > 0.95 and ags.min() > 0.9)
    assert all(checks) and loss_ok, (checks, loss_ok, r)


def test_multi_axis_bank_carry_over():
    """6 frames, 16x16 latents, 2 steps, alpha_t > 0, window 4 (two overlapping yt windows), VidToMe ON: per step the oracle is fed the
    latents the engine had and must reproduce the fused noise it handed to the scheduler, with the engine's maps injected.  The yt pass of
    a step replays maps whose `mrg2`/`bmap` refer to banks the xy pass left: if the engine had reset (or failed to carry) the banks between
    the passes the first yt chunk would be a seeding event without `mrg2` and the assertion on the trace structure below fails."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd.generate import Generator
    sd_unet, _, unet, _ = _engines(vidtome_seed=5)
    n, hh, ww = 6, 16, 16
    cfg = dict(n_timesteps=2, alpha_t=0.3, final_factor_t=0.5, win_size_t=4, chunk_size=4, guidance_scale=2.0, seed=77, noise_mode="vanilla")
    g = Generator(unet, None, cfg)
    g.prepare_data(torch.zeros(n, 3, 8 * hh, 8 * ww, device="cuda"))
    gen = torch.Generator().manual_seed(3)
    conds, conds_t = torch.randn(2, 154, 768, generator=gen).half(), torch.randn(2, 77, 768, generator=gen).half()
    cc = torch.randn(n, 4, hh, ww, generator=gen).half()
    x0 = g.init_noise.clone()
    rec = E.Recorder(g)
    x_hip = g.ddim_sample(x0.clone(), conds.cuda(), conds_t.cuda(), cc.cuda()).float().cpu()
    torch.cuda.synchronize()
    rec.finish()
    # trace structure: per merging block and step, exactly ONE seeding event (the first xy chunk); every yt chunk merges against a bank
    per_block = {}
    for t in rec.traces:
        per_block.setdefault(t["name"], []).append("mrg2" in t)
    assert len(per_block) == 10
    for name, ev in per_block.items():
        assert ev.count(False) == 2, (name, ev)                 # 2 steps -> 2 seeds; all later chunks (xy and yt) use the carried bank
    tome = E.InjectedToMe(rec.traces)
    fused = []
    with torch.no_grad():
        # replay step by step from the engine's own latents (per-step parity, no error carry-over)
        from oracle.scheduler import Scheduler as OSch
        from tc_light_amd import hostlogic as HL
        c = g.cfg
        osch = OSch(c.n_timesteps)
        alphas = E.OP.alpha_schedule(c.alpha_t, c.final_factor_t, c.n_timesteps)
        xy_s, yt_s = HL.ChunkSampler(c.seed, c.chunk_size, c.merge_global, c.chunk_ord), HL.ChunkSampler(c.seed + 1, c.chunk_size, c.merge_global, c.chunk_ord)
        ccf, text, text_t = cc.float(), conds.float(), conds_t.float()

        def pred(xin, txt, t, size):
            return E.OP.cfg(E.OS.unet_forward(sd_unet, torch.cat([xin, xin]), t, txt, tome.hook(size)), c.guidance_scale)
        for i, t in enumerate(osch.timesteps.tolist()):
            x = rec.seen[i][0]
            noises = torch.zeros_like(x)
            for ch in xy_s.get_chunks(n):
                noises[ch] = pred(torch.cat([x[ch], ccf[ch]], 1), text, float(t), (hh, ww))
            fz = E.OP.temporal_denoise(x, ccf, alphas[i], noises, c.win_size_t, [torch.as_tensor(ch) for ch in yt_s.get_chunks(ww)],
                                       lambda xt, ct, ch, sl: pred(torch.cat([xt, ct], 1), text_t, float(t), (xt.shape[2], hh)))[1]
            tome.reset()
            r = E.rel(rec.seen[i][1], fz)
            print(f"[multi-axis, VidToMe on, injected] step {i}: fused eps rel-L2 = {r:.3e}")
            fused.append(r)
    assert tome.exhausted()
    assert max(fused) < 1e-2, fused
    assert torch.isfinite(x_hip).all()
