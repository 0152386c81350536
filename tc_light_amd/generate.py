"""TC-Light pipeline driver on the MI355X engine -- host-side mirror of generate.py::Generator (reference).

  prepare_data -> _prepare_prompts_and_conditions -> ddim_sample{pred_noise per xy-chunk; temporal_denoise (yt-plane);
  scheduler.step} -> decode_latents_batch -> exposure_align -> unique_tensor_optimization        (generate.py:560-611)

Python sequences kernel launches and collectives; tensors never leave the device inside the timed region and there is no
per-iteration host sync (the reference syncs on loss.item() and moves the token bank through the CPU).
"""
import math
import os
import time
from types import SimpleNamespace

import numpy as np
import torch

from .lib import lib, stream
from . import hostlogic as HL
from . import post_opt
from .parallel import Dist, sharded_temporal_pass
from .scheduler import DPMSolverSDEScheduler

H16 = torch.float16
I32 = torch.int32

DEFAULTS = dict(  # configs/tclight_default.yaml (generation / post_opt sections)
    guidance_scale=2.0, n_timesteps=25, chunk_size=4, chunk_ord="mix-4", local_merge_ratio=0.6, merge_global=True,
    global_merge_ratio=0.5, global_rand=0.5, align_batch=True, max_downsample=2, noise_mode="same", alpha_t=0.0,
    final_factor_t=0.01, win_size_t=64, apply_opt=True, epochs_exposure=35, epochs=70, batch_size=16, lambda_dssim=0.2,
    lambda_flow=0.8, lambda_tv=0.05, feature_lr=0.05, exposure_lr_init=0.01, exposure_lr_final=0.001, seed=12345,
    post_opt_mode="replicated",      # multi-GPU only: how stage 1 / stage 2 run once the decoded frames are all-gathered.
                                     # "replicated" (default): every rank runs stage 2 in full on all frames, no collective at all -- it is
                                     #   bound by streams that do not shrink with the mini-batch share (Adam over the codebook rows, scatters), and
                                     #   path 2 is bit-reproducible, so the replicas agree bit for bit; stage 1, whose per-iteration work IS
                                     #   the mini-batch's slots and whose only exchange is the [N,3,4] gradient (14 KB all-reduce), is dealt over
                                     #   the ranks (round 5).  "replicated_all": stage 1 replicated too -- the one-GPU result, bit for bit.  "global": ONE parameter set with the mini-batch dealt over the ranks, gradients meet in
                                     #   all_reduce / reduce_scatter (2 x 12 K bytes per stage-2 iteration over xGMI: slower than replication
                                     #   from K ~ 1e7 up; kept for memory-bound cases).  "shard": the round-1 approximation, each rank's frame
                                     #   block as a video of its own (tracks cut at the block seams) -- NOT the reference's result.
    shard_post_opt=False,            # legacy spelling of post_opt_mode="shard"
    max_tokens_per_pass=None)
    # ^ None: TCL_MAX_TOKENS_PER_PASS (read when the Generator is built), else min(16 M, 80 % of the device's free memory / 10.5 KB) -- a block-major pass
    #   peaks at ~10.4 KB of activations per level-0 token (90 GB for the 8.64 M tokens of 300 frames at 1280x720): a 288 GB MI355X gets the 16 M
    #   below, a smaller part or a GPU shared with other processes a cap that fits.
    #   level-0 tokens (samples x pixels) one block-major UNet pass may carry; longer chunk lists are split into consecutive groups.  16 M = the
    #   whole xy step of 555 frames at 1280x720 in ONE pass (8.6 M tokens, 90 GB peak at 300 frames): rounds 1-3 used 1.5 M (int32-sized tensors);
    #   one group is 1.4 % faster (177.3 vs 179.8 s denoise, profiles/r4_ab_tokens_per_pass.txt) and, more important, its GEMM shapes do not
    #   depend on the random chunk lengths (groups cut at a cap hold 153-156 frames depending on the draws: every step met un-tabled shapes and
    #   the in-call tuner -- the 4.5 M line of that file).  tests/test_gpu_fullsize.py pins pass-size invariance (same bits) beyond 2^31 elements.


class Generator:
    def __init__(self, unet, vae, config=None, dist=None, scheduler=None, rmbg=None):
        cfg = dict(DEFAULTS)
        cfg.update(config or {})
        self.cfg = SimpleNamespace(**cfg)
        self.unet, self.vae, self.rmbg = unet, vae, rmbg
        self.dev = unet.dev
        self.L = lib()
        self.dist = dist or Dist()
        self.scheduler = scheduler or DPMSolverSDEScheduler()
        c = self.cfg
        # vidtome.apply_patch(...) (generate_utils.py:98-100); max_downsample is NOT forwarded by the reference (default 2)
        t = self.unet.tome
        t.args.update(local_merge_ratio=c.local_merge_ratio, merge_global=c.merge_global, global_merge_ratio=c.global_merge_ratio,
                      global_rand=c.global_rand, align_batch=bool(c.align_batch))
        self.batch_size = 2
        self.timing = {}
        if c.max_tokens_per_pass is None:
            env = os.environ.get("TCL_MAX_TOKENS_PER_PASS")
            if env:
                c.max_tokens_per_pass = int(env)
            else:
                free = 288 << 30
                if torch.cuda.is_available() and self.dev.type == "cuda":      # the driver's free bytes + what this process's caching allocator holds unused
                    free = torch.cuda.mem_get_info(self.dev)[0] + torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)
                cap = self.unet.panel_cache_cap() if hasattr(self.unet, "panel_cache_cap") else 0     # persistent attention panels live beside the pass
                cached = sum(a.numel() + b.numel() for a, b in getattr(self.unet, "_panel_cache", {}).values())
                free -= max(0, cap - cached)
                c.max_tokens_per_pass = int(min(16_000_000, max(1_000_000, 0.8 * free / 10_500)))

    # ------------------------------------------------------------------ data
    def prepare_data(self, frames, background=None):
        """frames: this rank's block [n_local,3,H,W] f32 in [0,1] on the device (generate.py:138-204).  background [1|n_local,3,H,W]:
        background compositing (generate.py:147-167): alpha from BriaRMBG on the resized frames, `alpha*fg + (1-alpha)*bg`."""
        c = self.cfg
        if background is not None:
            if self.rmbg is None:
                raise RuntimeError("background compositing needs an RMBGEngine (Generator(..., rmbg=...))")
            alpha = self.rmbg.estimate_alpha(frames, self.batch_size)
            frames = alpha * frames + (1 - alpha) * background.to(frames)
        self.frames = frames.contiguous()
        n, _, H, W = frames.shape
        self.h, self.w = H // 8, W // 8
        self.n_total = getattr(self, "n_total", None) or n * self.dist.world
        self.rng_dev = torch.Generator(device=self.dev).manual_seed(int(c.seed))
        if c.noise_mode.lower() == "same":        # prepare_latents(1, 4, H, W) repeated (generate.py:183-188)
            z = torch.randn(1, 4, self.h, self.w, generator=self.rng_dev, device=self.dev, dtype=torch.float32)
            self.init_noise = (z * self.scheduler.init_noise_sigma).to(H16).repeat(n, 1, 1, 1).contiguous()
        elif c.noise_mode.lower() == "vanilla":
            z = torch.randn(self.n_total, 4, self.h, self.w, generator=self.rng_dev, device=self.dev, dtype=torch.float32)
            lo, hi = self.dist.range(self.n_total)
            self.init_noise = (z[lo:hi] * self.scheduler.init_noise_sigma).to(H16).contiguous()
        else:
            raise NotImplementedError(f"Noise mode '{c.noise_mode}' is not supported.")

    # ------------------------------------------------------------------ one UNet evaluation with CFG
    def _groups(self, chunks, tokens_per_frame):
        """Consecutive chunks per UNet pass: as many as fit `max_tokens_per_pass` (activation memory); order is preserved, so every
        block still meets the chunks in the reference order."""
        cap = max(1, int(self.cfg.max_tokens_per_pass) // (2 * tokens_per_frame))
        groups, cur, n = [], [], 0
        for c in chunks:
            if cur and n + len(c) > cap:
                groups.append(cur); cur, n = [], 0
            cur.append(c); n += len(c)
        if cur:
            groups.append(cur)
        return groups

    def _run_groups(self, groups, Hh, Ww, t, text, pack, unpack):
        """The UNet passes of one list of chunk groups.  pack(grp) -> (xin, idx, n); unpack(eps, idx, n).  TCL_SKEW=1: the groups go through
        two at a time, half a transformer block apart (unet.py forward_pair) -- a lone group of several chunks is cut in two first; same bits."""
        mode = os.environ.get("TCL_SKEW", "0")                  # "cut": the same groups, one after the other (the A/B and parity partner of "1")
        if mode != "0":
            cut = []
            for grp in groups:
                if len(grp) < 2:
                    cut.append(grp)
                    continue
                tot = sum(len(c) for c in grp)
                heads = [sum(len(c) for c in grp[:k]) for k in range(1, len(grp))]
                k = 1 + min(range(len(heads)), key=lambda i: abs(2 * heads[i] - tot))         # the chunk boundary nearest to half the frames
                cut += [grp[:k], grp[k:]]
            groups = cut
            while mode != "cut" and len(groups) >= 2:
                ga, gb = groups[0], groups[1]
                groups = groups[2:]
                (xa, ia, na), (xb, ib, nb) = pack(ga), pack(gb)
                ea, eb = self.unet.forward_pair(xa, [len(c) for c in ga], xb, [len(c) for c in gb], Hh, Ww, t, text, cfg_pair=True)
                unpack(ea, ia, na); unpack(eb, ib, nb)
        for grp in groups:
            xin, idx, n = pack(grp)
            eps = self.unet.forward_many(xin, [len(c) for c in grp], Hh, Ww, t, text, cfg_pair=True)     # the pack kernel wrote both halves
            unpack(eps, idx, n)

    def _unet_xy(self, x, cc, chunks, text, t, noises):
        """pred_noise on the xy chunks of one step (generate.py:220-224, 288-352): chunks = lists of local frame ids, reference order.
        The chunks go through the UNet in block-major passes (`forward_many`, see unet.py): one pack, one unpack per pass."""
        L = self.L

        def pack(grp):
            frames = [f for c in grp for f in c]
            n = len(frames)
            idx = torch.tensor(frames, dtype=I32, device=self.dev)
            xin = torch.empty(2 * n, self.h, self.w, 8, dtype=H16, device=self.dev)
            L.tcl_pack_latents_f16(x, cc, idx, n, 0, 0, 0, self.h, self.w, xin, stream())
            return xin, idx, n

        def unpack(eps, idx, n):
            L.tcl_unpack_cfg_f16(eps, idx, n, 0, 0, 0, self.h, self.w, float(self.cfg.guidance_scale), 0, 1.0, 0, noises, stream())

        self._run_groups(self._groups(chunks, self.h * self.w), self.h, self.w, t, text, pack, unpack)

    def _unet_yt(self, x_full, cc_full, items, nt_full, text_t, t):
        """pred_noise on the yt chunks of one frame window: 'n c h w -> w c n h' (generate.py:265-273); items share (start, length)."""
        L = self.L
        sl, nwin, _, scale_upto, nkeep = items[0]

        def pack(grp):
            cols = [c for ch in grp for c in ch]
            n = len(cols)
            idx = torch.tensor(cols, dtype=I32, device=self.dev)
            xin = torch.empty(2 * n, nwin, self.h, 8, dtype=H16, device=self.dev)
            L.tcl_pack_latents_f16(x_full, cc_full, idx, n, 1, sl, nwin, self.h, self.w, xin, stream())
            return xin, idx, n

        def unpack(eps, idx, n):
            L.tcl_unpack_cfg_f16(eps, idx, n, 1, sl, nwin, self.h, self.w, float(self.cfg.guidance_scale), scale_upto, math.sqrt(0.5),
                                 nkeep, nt_full, stream())

        self._run_groups(self._groups([it[2] for it in items], nwin * self.h), nwin, self.h, t, text_t, pack, unpack)

    def _yt_items(self, w_chunks):
        """(window start, length, columns, scale_upto, nkeep) in the reference's loop order (generate.py:265-278)."""
        starts, ovl = HL.temporal_windows(self.n_total, self.cfg.win_size_t)
        items = []
        for k, sl in enumerate(starts):
            nwin = min(self.cfg.win_size_t, self.n_total - sl)
            nkeep = (starts[k + 1] - sl) if k + 1 < len(starts) else nwin
            up = sl + ovl[k - 1] if k > 0 else 0
            for ch in w_chunks:
                items.append((sl, nwin, ch, up, nkeep))
        return items

    # ------------------------------------------------------------------ hot loop 1
    @torch.no_grad()
    def ddim_sample(self, x, conds, conds_t, concat_conds):
        c, d = self.cfg, self.dist
        sch = self.scheduler
        sch.set_timesteps(c.n_timesteps)
        n_local = x.shape[0]
        noises = torch.zeros_like(x)
        noises_t = torch.zeros_like(x)
        alphas = HL.alpha_schedule(c.alpha_t, c.final_factor_t, len(sch.timesteps))
        xy_sampler = HL.ChunkSampler(c.seed + 7919 * d.rank, c.chunk_size, c.merge_global, c.chunk_ord)
        yt_sampler = HL.ChunkSampler(c.seed + 1, c.chunk_size, c.merge_global, c.chunk_ord)
        cc_full = d.gather_frames(concat_conds, self.n_total) if c.alpha_t > 0 else None
        lo, hi = d.range(self.n_total)
        for i, t in enumerate(sch.timesteps.tolist()):
            self._unet_xy(x, concat_conds, xy_sampler.get_chunks(n_local), conds, t, noises)
            if c.alpha_t > 0:
                items = self._yt_items(yt_sampler.get_chunks(self.w))
                nt = sharded_temporal_pass(d, x, cc_full, self.n_total, items,
                                           lambda xf, cf, its, out: self._unet_yt(xf, cf, its, out, conds_t, t))
                noises_t.copy_(nt)
                self.L.tcl_adain_fuse_f16(noises_t, noises, n_local * 4, self.h * self.w, float(alphas[i]), stream())
            z = None
            if i < len(sch.timesteps) - 1:     # the last SDE step has sigma_t = 0: its noise term vanishes
                zf = torch.randn(self.n_total, 4, self.h, self.w, generator=self.rng_dev, device=self.dev, dtype=torch.float32)
                z = zf[lo:hi].to(H16).contiguous()
            sch.step(noises, t, x, noise=z)
            self.unet.tome.reset_global_tokens()       # post_iter (generate_utils.py:235-238)
        return x

    # ------------------------------------------------------------------ end to end
    def __call__(self, frames, conds, conds_t, past_flows, mask_bwds, unq_inv, n_total=None, k=None, background=None):
        """frames: local block [n_local,3,H,W]; conds / conds_t: [2,L,768] f16 (uncond, cond) text embeddings for the xy / yt
        passes (generate.py:553-555); past_flows / mask_bwds / unq_inv: stage-2 inputs for ALL frames (device).
        Returns (relit frames [N,3,H,W] f32, info dict)."""
        c, d = self.cfg, self.dist
        self.n_total = n_total or frames.shape[0] * d.world
        ev = lambda: (torch.cuda.synchronize(self.dev), time.perf_counter())[1]
        t0 = ev()
        self.prepare_data(frames, background)
        concat_conds = self.vae.encode_imgs_batch(self.frames, self.batch_size)
        t1 = ev()
        x = self.ddim_sample(self.init_noise.clone(), conds, conds_t, concat_conds)
        t2 = ev()
        mode = "shard" if c.shard_post_opt else str(c.post_opt_mode)
        if mode not in ("replicated", "replicated_all", "global", "shard"):
            raise ValueError(f"post_opt_mode {mode!r}: expected replicated | replicated_all | global | shard")
        shard = d.multi and mode == "shard" and c.apply_opt
        lo, hi = d.range(self.n_total)
        if d.multi and not shard:
            # decode slab by slab; every slab goes into an async all-gather while the next one decodes (parallel.gather_frames_pipelined)
            clean_local = None
            clean = d.gather_frames_pipelined(lambda a, b: self.vae.decode_latents_batch(x[a:b], self.batch_size), x.shape[0], self.n_total,
                                              slab=int(os.environ.get("TCL_DECODE_SLAB", "8")))
        else:
            clean_local = self.vae.decode_latents_batch(x, self.batch_size)
        if shard:
            # Opt-in approximation: stage 1/2 on this rank's frame block as a video of its own (its first frame has no predecessor, tracks
            # are cut at the block boundary; the global track ids restricted to the block and renumbered densely give the partition
            # get_flowid would produce when started at the block's first frame).  No collective, but NOT the reference's result.
            clean = clean_local
            past_flows, mask_bwds = past_flows[lo:hi], mask_bwds[lo:hi]
            hw = clean_local.shape[-2] * clean_local.shape[-1]
            uniq, inv_local = torch.unique(unq_inv[lo * hw:hi * hw], return_inverse=True)
            unq_inv, k = inv_local.to(torch.int32), int(uniq.numel())
        elif clean_local is not None:
            clean = clean_local
        t3 = ev()
        losses1 = losses2 = None
        if c.apply_opt:
            N = clean.shape[0]
            pd = d if (mode == "global" and d.multi) else None      # global: the ranks split every mini-batch and share one parameter set;
            #                                                           # replicated / shard: every rank optimises on its own (no collective)
            pd1 = d if (mode in ("global", "replicated") and d.multi) else None      # stage 1: dealt over the ranks unless replicated_all / shard
            ds = post_opt.OptDataset(clean, past_flows, mask_bwds, device=self.dev)
            rng = np.random.default_rng(c.seed + (d.rank if shard else 0))      # global mode: the identical schedule on every rank
            s1 = post_opt.make_schedule(N, c.batch_size, c.epochs_exposure, rng)
            _, _, losses1 = post_opt.exposure_align(ds, s1, c.epochs_exposure, c.batch_size, c.exposure_lr_init, c.exposure_lr_final,
                                                    c.lambda_dssim, c.lambda_flow, dist=pd1)
            t4 = ev()
            if c.epochs > 0:
                s2 = post_opt.make_schedule(N, c.batch_size, c.epochs, rng)
                clean, _, losses2 = post_opt.unique_tensor_optimization(ds, unq_inv, s2, c.batch_size, c.feature_lr, c.lambda_dssim,
                                                                        c.lambda_flow, c.lambda_tv, k=k, dist=pd)
            else:
                clean = ds.edited_images
            if shard:
                clean = d.gather_frames(clean.contiguous(), self.n_total)
            t5 = ev()
        else:
            t4 = t5 = t3
        self.timing = dict(encode=t1 - t0, denoise=t2 - t1, decode=t3 - t2, stage1=t4 - t3, stage2=t5 - t4, total=t5 - t0)
        info = dict(timing=self.timing, losses_exposure=losses1, losses_unique=losses2, total_time=t5 - t0,
                    sec_per_frame=(t5 - t0) / self.n_total, total_number_of_frames=self.n_total,
                    max_memory_allocated=torch.cuda.max_memory_allocated(self.dev) / 1024.0 ** 2)
        return clean, info
