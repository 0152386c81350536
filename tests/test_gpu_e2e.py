"""GPU: END-TO-END parity of the whole path -- Generator.__call__ (reference generate.py:560-611: VAE encode -> denoising loop with VidToMe
ON through the default block-major `forward_many` schedule -> VAE decode -> stage 1 -> stage 2) against the same composition on the CPU
oracle (tests/e2e_oracle.py) with identical seeds: same synthetic frames, seeded weights, initial noise, chunk draws, VidToMe draws, SDE
noise and mini-batch schedules.

test_config1_*           = BASELINE.json configs[0] IN FULL: 8 frames 512x512, 4 denoising steps, single axis (alpha_t = 0), VidToMe 0.6/0.5,
                           stage 1 35 epochs + stage 2 70 epochs (TCL_E2E_EPOCHS="a,b" shortens the optimiser for quick local runs).  One engine
                           run (the launch test, first in the session) and the oracle legs as background processes (tests/e2e_jobs.py), collected
                           by three tests at the end of the session (round 6: the ~430 s of CPU oracle no longer sit inside one test).
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


# ---------------------------------------------------------------------------------------------------------------- configs[0] in full
# Round 6: ONE engine run, recorded; the oracle legs (f32 denoise + VAE, its f16 floor, stage 1/2 three ways, the oracle's own matching) run as
# background processes on the host's cores (tests/e2e_jobs.py) while the GPU goes through the rest of the suite; the tests below collect them.
_S = {}
# share of the CPU budget per leg.  `floor` (a second whole-path oracle denoise with f16 op outputs) runs only with TCL_E2E_FULL=1: its figures have been
# the same to three digits in every run since round 5 (FLOOR below), and the chunk-level floor is measured live in tests/test_gpu_unet.py.
_LEGS = dict(denoise=0.4, computed=0.2, post_same=0.2, post_pert=0.2)
_FULL = os.environ.get("TCL_E2E_FULL", "0") != "0"
FLOOR = dict(encode=1.87e-3, latents=1.60e-3, decoded=9.68e-4)       # oracle with f16 op outputs vs f32 oracle, configs[0]: profiles/r6_e2e_legs_full.log


def cpu_budget():
    """CPUs this process may really use: the GPU box's container shows 256 logical CPUs but runs under a cgroup quota (measured round 6: ~15 busy cores
    however many threads ask) -- oversubscribing it made every oracle leg 5-9x slower AND starved the GPU tests' own host threads."""
    n = os.cpu_count() or 8
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            elif int(txt[0]) > 0:
                n = min(n, max(1, int(int(txt[0]) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
        except (OSError, ValueError, IndexError):
            pass
    return int(os.environ.get("TCL_E2E_BUDGET", max(4, min(n, 16) - 4)))      # (16: what the legs were measured to use well; 4 stay with the GPU tests)


def _launch():
    if _S:
        return _S
    import atexit
    import subprocess
    import sys
    import tempfile
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
    nc = int(os.environ.get("TCL_E2E_COMPUTED", "2"))
    wd = tempfile.mkdtemp(prefix="tcl_e2e_")
    torch.save(dict(n=n, H=H, W=W, cfg=vars(gen.cfg), x0=gen.init_noise.float().cpu(), zs=rec.zs, traces=rec.traces, draws=rec.draws,
                    clean_engine=stages["clean"].float().cpu(), nc=nc, full=_FULL), os.path.join(wd, "state.pt"))
    budget = cpu_budget()
    legs = dict(_LEGS, floor=0.4) if _FULL else dict(_LEGS)
    if nc <= 0:
        legs.pop("computed")
    tot = sum(legs.values())
    procs = {}
    for leg, share in legs.items():
        th = max(2, int(round(share / tot * budget)))
        env = dict(os.environ, OMP_NUM_THREADS=str(th), MKL_NUM_THREADS=str(th), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        log = open(os.path.join(wd, leg + ".log"), "w")
        procs[leg] = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_jobs.py"), leg, wd, str(th)],
                                      env=env, stdout=log, stderr=subprocess.STDOUT)
    atexit.register(lambda: [p.kill() for p in procs.values() if p.poll() is None])
    _S.update(wd=wd, procs=procs, budget=budget, t_launch=time.time(), t_hip=t_hip, n=n, inv=inv, d=d, nc=nc, cfg=gen.cfg,
              out=out.float().cpu(), l1h=info["losses_exposure"].cpu().numpy(), l2h=info["losses_unique"].cpu().numpy(),
              stages={k_: v.float().cpu() for k_, v in stages.items()}, n_zs=len(rec.zs), n_traces=len(rec.traces), n_draws=len(rec.draws),
              lat_after=[s_[0] for s_ in rec.seen])
    return _S


def _collect(name, leg):
    """Wait for a leg's output file (the leg normally finished long ago: it ran beside the other GPU tests).  TCL_E2E_WAIT bounds the wait."""
    S = _launch()
    path, p = os.path.join(S["wd"], name), S["procs"][leg]
    limit = time.time() + float(os.environ.get("TCL_E2E_WAIT", "140"))
    while not os.path.exists(path):
        if p.poll() is not None and not os.path.exists(path):
            raise AssertionError(f"oracle leg {leg!r} ended with rc {p.returncode} without {name}:\n" + open(os.path.join(S["wd"], leg + ".log")).read()[-3000:])
        if time.time() > limit:
            raise AssertionError(f"oracle leg {leg!r} did not produce {name} within the wait ({time.time() - S['t_launch']:.0f} s since launch)")
        time.sleep(0.5)
    r = torch.load(path, weights_only=False)
    print(f"[e2e config 1] oracle leg {leg} -> {name}: {r['seconds']:.0f} s on {r['threads']} threads (in the background; {time.time() - S['t_launch']:.0f} s since launch)")
    return r


def test_config1_launch_engine_run_and_oracle_legs():
    """BASELINE.json configs[0] IN FULL on the engine (Generator.__call__, generate.py:560-611), recorded; starts the oracle legs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    S = _launch()
    assert torch.isfinite(S["out"]).all() and S["n_zs"] == 4
    assert S["n_traces"] == 10 * S["n_draws"] and S["n_draws"] >= 4 * 2           # 10 merging blocks x chunks x steps
    print(f"[e2e config 1] engine run {S['t_hip']:.1f} s (cold); oracle legs started in {S['wd']}: " + ", ".join(S["procs"]))


def test_config1_denoise_decode_vs_oracle():
    """VAE encode -> denoising loop (VidToMe ON, engine's maps injected) -> VAE decode against the f32 oracle: north_star's 1e-3 on the relit frames
    out of denoise + decode, and every stage against the f16 noise floor of the same composition."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    S = _launch()
    o = _collect("out_denoise.pt", "denoise")
    st = S["stages"]
    floor = dict(FLOOR)
    if _FULL:
        f = _collect("out_floor.pt", "floor")
        floor = dict(encode=E.rel(f["cc"], o["cc"]), latents=E.rel(f["lat"], o["lat"]), decoded=E.rel(f["clean"], o["clean"]))
        assert all(abs(floor[k_] / FLOOR[k_] - 1) < 0.05 for k_ in FLOOR), (floor, FLOOR)          # the recorded constants still hold
    r = dict(encode=E.rel(st["cc"], o["cc"]), latents=E.rel(st["lat"], o["lat"]), decoded=E.rel(st["clean"], o["clean"]))
    print("[e2e config 1] f16 noise floor (oracle with f16 op outputs vs f32 oracle" + (", measured in this run" if _FULL else ", recorded: TCL_E2E_FULL=1 measures it")
          + "): " + ", ".join(f"{k_} {v:.2e}" for k_, v in floor.items()))
    print("[e2e config 1, injected maps] engine vs f32 oracle rel-L2: " + ", ".join(f"{k_} {v:.2e}" for k_, v in r.items()))
    assert r["encode"] < 2e-3 and r["latents"] < 5e-3, r
    assert r["decoded"] < 1.0e-3, r             # north_star's 1e-3 rel-L2 on the relit frames out of the denoise + decode path
    for k_ in r:                                # ... and no further from f32 than an op-by-op f16 pipeline is
        assert r[k_] < 1.25 * floor[k_], (k_, r, floor)


def test_config1_post_opt_vs_oracle():
    """Stage 1 (35 epochs) + stage 2 (70 epochs).  Adam's update is +-lr whatever the gradient's size, so rounding noise decides the sign wherever the
    gradient nearly cancels and 105 iterations spread that: the oracle run twice with inputs 1e-7 apart ends ~1.2e-2 apart (measured here) -- no
    pointwise 1e-3 exists for anyone after stage 2.  Asserted: the engine within 1.5x the oracle's own self-distance (from the same decoded frames
    and over the whole path), EVERY iteration's loss to 2e-4 (same decoded frames) / 2e-3 (whole path), and the statistics that are not chaotic."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    S = _launch()
    same, pert = _collect("out_post_same.pt", "post_same"), _collect("out_post_pert.pt", "post_pert")
    out, final_h, final_p = S["out"], same["final"], pert["final"]
    l1h, l2h = S["l1h"], S["l2h"]
    l1s, l2s = np.asarray(same["l1"]), np.asarray(same["l2"])
    r = dict(final_same_decoded=E.rel(out, final_h), oracle_vs_oracle_1e7=E.rel(final_p, final_h))
    print(f"[e2e config 1] stage-1 loss first/last HIP {l1h[0]:.5f}/{l1h[-1]:.5f} oracle {l1s[0]:.5f}/{l1s[-1]:.5f}; "
          f"stage-2 HIP {l2h[0]:.5f}/{l2h[-1]:.5f} oracle {l2s[0]:.5f}/{l2s[-1]:.5f}")
    dd = (out - final_h).abs()
    print(f"[e2e config 1] final vs oracle-from-same-decoded: median |diff| {dd.median().item():.2e}, fraction > 1e-2: {(dd > 1e-2).float().mean().item():.4f}")
    self_d = max(r["oracle_vs_oracle_1e7"], 2e-3)
    print(f"[e2e config 1] loss trajectories, max relative difference, same decoded frames: stage 1 {np.abs(l1h / l1s - 1).max():.2e} stage 2 {np.abs(l2h / l2s - 1).max():.2e}")
    assert np.allclose(l1h, l1s, rtol=2e-4) and np.allclose(l2h, l2s, rtol=2e-4)
    if _FULL:           # ... and over the WHOLE path: stage 1 + 2 on the oracle from the ORACLE's decoded frames (1e-3 from the engine's)
        whole = _collect("out_whole.pt", "denoise")
        l1, l2 = np.asarray(whole["l1"]), np.asarray(whole["l2"])
        r["final"] = E.rel(out, whole["final"])
        print(f"[e2e config 1] whole path: loss trajectories max relative difference {np.abs(l1h / l1 - 1).max():.2e} / {np.abs(l2h / l2 - 1).max():.2e}")
        assert np.allclose(l1h, l1, rtol=2e-3) and np.allclose(l2h, l2, rtol=2e-3)
        assert r["final"] < 1.5 * self_d, r
    print("[e2e config 1] after stage 1 + 2, rel-L2: " + ", ".join(f"{k_} {v:.2e}" for k_, v in r.items()))
    assert r["final_same_decoded"] < 1.5 * self_d, r
    # statistics of the FINAL frames that are not chaotic (sign noise of single codebook rows averages out): per-frame mean colour, mean colour of
    # 64 groups of tracks (track id mod 64), the masked warp error of the result.  Scale: the oracle's own distance between its two runs.
    fl, mk, inv = S["d"]["past_flows"], S["d"]["masks"], S["inv"]

    def stats(img):
        fm = img.mean(dim=(2, 3))
        flat = img.permute(0, 2, 3, 1).reshape(-1, 3)
        grp = (inv.long() % 64)
        gm = torch.zeros(64, 3).index_add_(0, grp, flat) / torch.bincount(grp, minlength=64).clamp_min(1)[:, None]
        warped = E.O2.warp_flow(img[:-1], fl[1:])
        we = ((warped - img[1:]).abs() * mk[1:]).mean()
        return fm, gm, we
    sh, so, sp = stats(out), stats(final_h), stats(final_p)
    d_fm, d_gm, d_we = (sh[0] - so[0]).abs().max().item(), (sh[1] - so[1]).abs().max().item(), abs(sh[2] / so[2] - 1).item()
    s_fm, s_gm, s_we = (sp[0] - so[0]).abs().max().item(), (sp[1] - so[1]).abs().max().item(), abs(sp[2] / so[2] - 1).item()
    print(f"[e2e config 1] non-chaotic statistics, engine vs oracle from the same decoded frames (oracle vs itself 1e-7 apart): per-frame mean colour "
          f"{d_fm:.2e} ({s_fm:.2e}), track-group mean colour {d_gm:.2e} ({s_gm:.2e}), masked warp error rel {d_we:.2e} ({s_we:.2e})")
    assert d_fm < max(3 * s_fm, 3e-4) and d_gm < max(3 * s_gm, 3e-4) and d_we < max(3 * s_we, 3e-3)


def test_config1_oracle_decides_its_own_matches():
    """The first TCL_E2E_COMPUTED (default 2) of the 4 steps with the oracle's own matching (f16-emulating score rule, same (randf, coin) draws):
    discrete decisions differ on near-tied f16 scores; asserted on provenance agreement and loosely on the latents."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    S = _launch()
    nc = S["nc"]
    if nc <= 0:
        pytest.skip("TCL_E2E_COMPUTED=0")
    c = _collect("out_computed.pt", "computed")
    lat_h = S["stages"]["lat"] if nc >= 4 else S["lat_after"][nc]              # the engine's latents after nc steps
    rc = dict(latents=E.rel(lat_h, c["lat"]))
    if "clean" in c:
        rc["decoded"] = E.rel(S["stages"]["clean"], c["clean"])
    ag, ags = np.asarray(c["agree"]), np.asarray(c["agree_src"])
    print(f"[e2e config 1, computed maps, {nc} steps] positions restored from the SAME source token: mean {ags.mean():.3f} min {ags.min():.3f} over {len(ags)} "
          f"merges (raw equality of the stored unmerge maps, whose slot numbering of unmerged tokens differs by design: mean {ag.mean():.3f}); rel-L2: "
          + ", ".join(f"{k_} {v:.2e}" for k_, v in rc.items()))
    # measured (round 5): 0.979 mean / 0.932 min over the 60 merges of 2 steps, on random-weight (near-isotropic) activations
    assert rc["latents"] < 2e-2 and ags.mean() > 0.95 and ags.min() > 0.9, rc


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
