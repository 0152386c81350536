"""Test harness (imports oracle/): the CPU-oracle legs of tests/test_gpu_e2e.py as BACKGROUND PROCESSES.

Round 5's GPU suite outgrew the driver's clock because `test_config1_end_to_end` ran ~430 s of CPU oracle serially inside one test, in front of
every leaf parity test.  Nothing about the oracle legs needs the GPU or each other: once the engine's run is recorded (initial
noise, SDE noise, chunk / VidToMe draws, merge maps, decoded frames -- `state.pt`) each leg is a pure function of that file.  The launcher test
(first in the session, tests/conftest.py orders it) writes the file and starts the legs as `python tests/e2e_jobs.py <leg> <dir> <threads>`;
they run on the host's cores while the GPU goes through the rest of the suite, and the collecting tests (last in the session) read what they
wrote.  Same oracle, same seeds, same assertions as round 5 -- only the wall-clock is shared.

Legs:
  denoise   f32 oracle: VAE encode -> denoising loop with the engine's merge maps injected -> VAE decode  (out_denoise.pt), then stage 1 + 2
            from the ORACLE's decoded frames (out_whole.pt)
  floor     the same composition with every op's output rounded to f16 (oracle/sd15.py `half_outputs`): the f16 noise floor of the path
  post_same stage 1 + 2 on the oracle from the ENGINE's decoded frames
  post_pert ... from the engine's decoded frames perturbed by 1e-7 relative: the oracle against itself (how chaotic 105 Adam steps are)
  computed  the oracle deciding its own matches for the first `nc` steps
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _inputs(st):
    import numpy as np
    import torch
    import synth
    from tc_light_amd import sd15
    from types import SimpleNamespace
    sd_unet = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    sd_vae = sd15.random_state_dict(sd15.vae_param_shapes(), seed=2)
    d = synth.video_clip(st["n"], st["H"], st["W"], seed=12345)
    inv, _ = synth.track_ids(st["n"], st["H"], st["W"], seed=3)
    g5, g6 = np.random.default_rng(5), np.random.default_rng(6)
    conds = torch.from_numpy(g5.standard_normal((2, 154, 768)).astype(np.float32)).half().float()
    conds_t = torch.from_numpy(g6.standard_normal((2, 77, 768)).astype(np.float32)).half().float()
    return sd_unet, sd_vae, d, inv, conds, conds_t, SimpleNamespace(**st["cfg"])


def main(leg, wd, threads):
    import torch
    torch.set_num_threads(int(threads))
    import e2e_oracle as E
    st = torch.load(os.path.join(wd, "state.pt"), weights_only=False)
    sd_unet, sd_vae, d, inv, conds, conds_t, c = _inputs(st)
    n = st["n"]
    t0 = time.time()

    def save(name, **kw):
        kw["seconds"] = time.time() - t0
        kw["threads"] = int(threads)
        tmp = os.path.join(wd, name + ".tmp")
        torch.save(kw, tmp)
        os.replace(tmp, os.path.join(wd, name))          # atomic: a reader never sees half a file

    if leg in ("denoise", "floor"):
        import contextlib
        ctx = E.OS.half_outputs() if leg == "floor" else contextlib.nullcontext()
        with ctx, torch.no_grad():
            cc = E.vae_batches(E.OS.vae_encode, sd_vae, d["frames"])
            tome = E.InjectedToMe(st["traces"])
            lat = E.oracle_denoise(sd_unet, st["x0"], cc, conds, conds_t, c, tome, st["zs"], c.seed, c.seed + 1)
            assert tome.exhausted()
            clean = E.vae_batches(E.OS.vae_decode, sd_vae, lat)
        save(f"out_{leg}.pt", cc=cc, lat=lat, clean=clean)
        if leg == "denoise" and st.get("full"):
            _, final_i, l1, l2 = E.oracle_post_opt(clean, d["past_flows"], d["masks"], inv, c, n)
            save("out_whole.pt", final=final_i, l1=[float(v) for v in l1], l2=[float(v) for v in l2])
    elif leg in ("post_same", "post_pert"):
        clean = st["clean_engine"]
        if leg == "post_pert":
            g7 = torch.Generator().manual_seed(1)
            clean = (clean * (1 + 1e-7 * torch.randn(clean.shape, generator=g7))).clamp(0, 1)
        _, final, l1, l2 = E.oracle_post_opt(clean, d["past_flows"], d["masks"], inv, c, n)
        save(f"out_{leg}.pt", final=final, l1=[float(v) for v in l1], l2=[float(v) for v in l2])
    elif leg == "computed":
        nc = st["nc"]
        with torch.no_grad():
            cc = E.vae_batches(E.OS.vae_encode, sd_vae, d["frames"])
            tome_c = E.ComputedToMe(st["draws"], traces=st["traces"])
            lat_c = E.oracle_denoise(sd_unet, st["x0"], cc, conds, conds_t, c, tome_c, st["zs"], c.seed, c.seed + 1, max_steps=nc)
            out = dict(lat=lat_c, agree=tome_c.agree, agree_src=tome_c.agree_src)
            if nc >= c.n_timesteps:
                out["clean"] = E.vae_batches(E.OS.vae_decode, sd_vae, lat_c)
        save("out_computed.pt", **out)
    else:
        raise SystemExit(f"unknown leg {leg!r}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
