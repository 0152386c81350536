"""GPU parity of the UNet engine (HIP, f16) against the fp32 CPU oracle on identical seeded weights and inputs.

Stage (i)  -- VidToMe indices injected from the HIP run into the oracle: pure numeric parity of ~700 chained kernels.
              Tolerance: rel-L2 <= 2.5e-3 on eps AND <= 1.25x the f16 noise floor (the oracle with f16 op outputs vs itself in f32,
              measured in the test: 1.7e-3; engine 1.6e-3).
Stage (ii) -- oracle computes its own matching with the f16-emulating rule: fraction of positions restored from the same source token
              (>= 0.98 per block asserted; measured 0.990-1.000) and rel-L2.
The oracle's UNet arithmetic itself is parity-UNPINNED w.r.t. diffusers (see oracle/sd15.py header).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd import sd15
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    assert sum(v.numel() for v in sd.values()) == 859_520_964 + 320 * 4 * 9      # SD-1.5 UNet + 4 extra conv_in channels
    tome = VidToMe("cuda", seed=5)
    eng = UNetEngine(sd, "cuda", tome)
    return sd, eng, tome


def _inputs(F, Hh, Ww, seed):
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.standard_normal((F, 8, Hh, Ww)).astype(np.float32)).half().float()
    text = torch.from_numpy(g.standard_normal((2, 77, 768)).astype(np.float32)).half().float()
    return x, text


def _run_hip(eng, x, text_dev, F, Hh, Ww, t):
    xin = torch.cat([x, x]).permute(0, 2, 3, 1).contiguous().cuda().half()        # cat([x,x]) (generate.py:298), NHWC
    eps = eng.forward_nhwc(xin, F, Hh, Ww, t, text_dev)
    torch.cuda.synchronize()
    return eps.view(2 * F, Hh, Ww, 4).permute(0, 3, 1, 2).float().cpu()


def test_unet_two_chunks(setup):
    from oracle import sd15 as OS
    from oracle import vidtome as OV
    sd, eng, tome = setup
    F, Hh, Ww, t = 2, 16, 24, 801.0
    text = _inputs(F, Hh, Ww, 0)[1]
    text_dev = text.cuda().half()
    tome.reset_global_tokens()
    tome.trace = []
    tome.draws = [(1, 0.9), (0, 0.2)]              # chunk 0 seeds the banks; chunk 1: coin 0.2 <= 0.5 -> bank tokens are src
    xs = [_inputs(F, Hh, Ww, 10 + k)[0] for k in range(2)]
    hip = [_run_hip(eng, xs[k], text_dev, F, Hh, Ww, t) for k in range(2)]
    traces = tome.trace
    tome.trace = None
    assert len(traces) == 20                         # 10 merging blocks (ds 1, 2) x 2 chunks
    # ---- stage (i): inject the HIP maps into the oracle
    banks32, banks16 = {}, {}
    banks = banks32
    it = iter(traces)

    def tome_injected(p, n1):
        B2, N, C = n1.shape
        if int(np.ceil(np.sqrt((Hh * Ww) // N))) > 2:
            return n1, (lambda y: y)
        tr = next(it)
        assert tr["name"] == p
        xj = n1.reshape(2, F * N, C)
        local = xj[:, tr["mrg1"].cpu().long()] if tr.get("mrg1") is not None else (xj[:, tr["gather"].cpu().long()] if tr.get("gather") is not None else xj)
        unm = tr["unm"].cpu().long() if tr["unm"] is not None else torch.arange(F * N)
        if "mrg2" not in tr:
            banks[p] = local.clone()
            return local, (lambda y: y[:, unm].reshape(B2, N, C))
        bank = banks[p]
        TL, Tb = tr["TL"], bank.shape[1]
        cat = torch.empty(2, TL + Tb, C)
        cat[:, tr["loff"]:tr["loff"] + TL] = local
        cat[:, tr["boff"]:tr["boff"] + Tb] = bank
        merged = cat[:, tr["mrg2"].cpu().long()]
        return merged, (lambda y: y[:, unm].reshape(B2, N, C))
    for k in range(2):
        it = iter(traces[10 * k:])
        ref = OS.unet_forward(sd, torch.cat([xs[k], xs[k]]), t, text, tome_injected)
        r = rel(hip[k], ref)
        # the f16 noise floor: the SAME oracle with every op's output rounded to f16 (the reference's torch.float16 pipeline, oracle/sd15.py
        # half_outputs) against the f32 oracle, same injected maps; each precision keeps its own banks from chunk 0 to chunk 1
        banks, it = banks16, iter(traces[10 * k:])
        with OS.half_outputs():
            ref16 = OS.unet_forward(sd, torch.cat([xs[k], xs[k]]), t, text, tome_injected)
        banks = banks32
        floor = rel(ref16, ref)
        print(f"[unet parity, injected indices] chunk {k}: engine vs f32 oracle rel-L2 = {r:.3e}; f16 noise floor (oracle f16 outputs vs f32) = {floor:.3e}")
        assert r < 2.5e-3 and r < 1.25 * floor, (r, floor)       # measured 1.6e-3 engine / 1.7e-3 floor: the engine is AT the floor of f16 arithmetic
        if k == 0:   # banks for chunk 1 in the oracle = unmerged local tokens; with no bank yet that is `local`
            pass
    # ---- stage (ii): oracle's own matching (f16-emulating tie rule), chunk 0 only (no bank dependence)
    import e2e_oracle as E
    agree, agree_src = [], []

    def tome_own(p, n1):
        B2, N, C = n1.shape
        if int(np.ceil(np.sqrt((Hh * Ww) // N))) > 2:
            return n1, (lambda y: y)
        r = OV.compute_merge(n1, F, None, 1, 0.9, emulate_f16=True)
        tr = next(t0)
        agree.append((r["unm"] == tr["unm"].cpu().long()).float().mean().item())
        ids = torch.arange(F * N)
        agree_src.append((E.oracle_provenance(r, ids, None)[0] == E.trace_provenance(tr, ids, None)[0]).float().mean().item())
        return r["merged"], r["unmerge"]
    t0 = iter(traces[:10])
    ref = OS.unet_forward(sd, torch.cat([xs[0], xs[0]]), t, text, tome_own)
    r = rel(hip[0], ref)
    print(f"[unet parity, computed indices] positions restored from the same source token, per block: {['%.3f' % a for a in agree_src]} "
          f"(raw equality of the stored maps -- unmerged slots are numbered differently by design: {['%.3f' % a for a in agree]}); rel-L2 = {r:.3e}")
    # random-weight activations are close to isotropic noise, so many cosine scores sit within one f16 ulp of each other and
    # the 1e-3-level activation differences flip some matches; the maps still mostly agree and the output stays close.
    assert min(agree_src) > 0.98 and min(agree) > 0.7 and r < 3e-2      # measured: 99.0-100 % of the positions per block restored from the same source
    #                                                                      token (raw map equality 80 %), eps rel-L2 1.2e-2


def test_forward_many_equals_sequential(setup):
    """forward_many (all chunks in one block-major pass, only attn1 of the merging blocks chunk by chunk) against the plain per-chunk
    loop of the reference (generate.py:220-224) with the same VidToMe draws: same bank chains per block, same chunk order.  Every kernel is
    deterministic (the per-chunk loop run twice is bit-identical -- asserted), but the two schedules are not bit-identical to each other: the
    K-split count of a GEMM and the block partition of the GroupNorm sums depend on the row count, so f32 sums are associated differently,
    and with random weights on a 16x24 latent a last-bit change flips near-tied matches.  Bounds: >= 85 % of all unmerge-map entries equal
    on average, eps within 2e-2 rel-L2 (measured 0.91 / 0.009)."""
    sd, eng, tome = setup
    Hh, Ww, t = 16, 24, 801.0
    Fs = [1, 3, 2]
    text_dev = _inputs(1, Hh, Ww, 0)[1].cuda().half()
    xs = [torch.cat([x, x]).permute(0, 2, 3, 1).contiguous().cuda().half() for x in (_inputs(F, Hh, Ww, 20 + k)[0] for k, F in enumerate(Fs))]
    draws = [(0, 0.9), (2, 0.3), (1, 0.7)]

    def run(many):
        tome.reset_global_tokens(); tome.draws = list(draws); tome.trace = []
        if many:
            Ft = sum(Fs)
            xa = torch.cat([x[:F] for x, F in zip(xs, Fs)] + [x[F:] for x, F in zip(xs, Fs)])        # uncond of all chunks, then cond
            ea = eng.forward_many(xa, Fs, Hh, Ww, t, text_dev).view(2 * Ft, -1)
            out, off = [], 0
            for F in Fs:
                out.append(torch.cat([ea[off:off + F], ea[Ft + off:Ft + off + F]]).reshape(-1, 4)); off += F
        else:
            out = [eng.forward_nhwc(x, F, Hh, Ww, t, text_dev).clone().reshape(-1, 4) for x, F in zip(xs, Fs)]
        tr = tome.trace
        tome.trace = None; tome.draws = None
        torch.cuda.synchronize()
        return out, tr

    def compare(ra, rb):
        (oa, ta), (ob, tb) = ra, rb
        assert len(ta) == len(tb) == 30
        by_block = lambda tr: {n: [d for d in tr if d["name"] == n] for n in {d["name"] for d in tr}}
        a, b = by_block(ta), by_block(tb)
        agree = []
        for n in a:                                      # per block, chunks arrive in the same order
            for da, db in zip(a[n], b[n]):
                assert da["T"] == db["T"]
                if da["unm"] is not None:
                    agree.append((da["unm"] == db["unm"]).float().mean().item())
        return min(agree), sum(agree) / len(agree), max(rel(y, x) for x, y in zip(oa, ob))

    seq1, seq2, many = run(False), run(False), run(True)
    base, got = compare(seq1, seq2), compare(seq1, many)
    print("sequential vs sequential (min/mean map agreement, max eps rel-L2):", base, " sequential vs forward_many:", got)
    assert base == (1.0, 1.0, 0.0)                # deterministic kernels: no float atomics on the UNet path
    assert got[1] > 0.86 and got[2] < 1.8e-2       # measured 0.91 / 0.009 (2x on the error, 1.5x on the disagreement)


@pytest.mark.parametrize("Hh,Ww,Fs,Lt", [(4, 8, [2, 4, 2], 77), (5, 9, [3], 77), (9, 5, [3], 154), (6, 10, [1, 2], 77)])
def test_unet_small_odd_planes_vs_oracle(setup, Hh, Ww, Fs, Lt):
    """Plane geometries the yt axis produces for short windows (rows = frames of the window): sizes that halve to 1 along one axis
    before the other, odd sizes whose up-sampling is not 2x (5 -> 3 -> 5), several chunks per pass.  VidToMe off (pure geometry).
    (4, 8) is the case that exposed the one-axis nearest up-sampling bug of the implicit conv (1x1 -> 1x2)."""
    from oracle import sd15 as OS
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    sd, _, _ = setup
    eng = UNetEngine(sd, "cuda", VidToMe("cuda", seed=5, enabled=False))
    F = sum(Fs)
    x, _ = _inputs(F, Hh, Ww, 40 + Hh)
    text = torch.from_numpy(np.random.default_rng(Hh).standard_normal((2, Lt, 768)).astype(np.float32)).half().float()
    xin = torch.cat([x, x]).permute(0, 2, 3, 1).contiguous().cuda().half()
    eps = eng.forward_many(xin, Fs, Hh, Ww, 501.0, text.cuda().half()).view(2 * F, Hh, Ww, 4).permute(0, 3, 1, 2).float().cpu()
    off = 0
    for f in Fs:
        ref = OS.unet_forward(sd, torch.cat([x[off:off + f], x[off:off + f]]), 501.0, text)
        got = torch.cat([eps[off:off + f], eps[F + off:F + off + f]])
        assert rel(got, ref) < 1e-2, (Hh, Ww, f, rel(got, ref))
        off += f


@pytest.mark.parametrize("vidtome_on,align", [(True, True), (True, False), (False, True)])
def test_cfg_pair_dedup_bit_identical(setup, monkeypatch, vidtome_on, align):
    """The classifier-free-guidance halves of a UNet call are identical until the first text cross-attention (generate.py:342-347: `cat([latents] * 2)`,
    the same concat_conds).  With cfg_pair=True the engine computes conv_in, the first ResNet block and proj_in / norm1 / VidToMe merge / attn1 of the
    first transformer block ONCE and duplicates them; the noise prediction must be THE SAME BITS as computing both halves (TCL_CFG_DEDUP=0), and the
    merge maps / banks of every block the same: every kernel on that prefix is deterministic and independent of the batch size (GroupNorm's row partition,
    the attention kernel variant via the pair flag; GEMM K splits are 1 from 16 384 rows up -- the test is sized above that)."""
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    sd = setup[0]
    tome = VidToMe("cuda", seed=5, enabled=vidtome_on, align_batch=align)
    eng = UNetEngine(sd, "cuda", tome)
    Hh, Ww, t = 32, 48, 601.0
    Fs = [4, 4, 3, 1]                                     # 12 frames x 1536 tokens = 18 432 rows in the half batch
    Ft = sum(Fs)
    g = torch.Generator(device="cuda").manual_seed(77)
    xh = torch.randn(Ft, Hh, Ww, 8, device="cuda", generator=g).half()
    x = torch.cat([xh, xh]).contiguous()
    text = torch.randn(2, 77, 768, device="cuda", generator=g).half()
    draws = [(1, 0.9), (3, 0.2), (0, 0.7), (0, 0.4)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TCL_CFG_DEDUP", mode)
        outs = []
        for step in range(2):                             # second call: every block's bank exists (bank-src and local-src merges)
            tome.draws = list(draws) if vidtome_on else None
            tome.trace = []
            eng.count_flops, eng.flops, eng.flops_executed = True, 0.0, 0.0
            eps = eng.forward_many(x, Fs, Hh, Ww, t, text, cfg_pair=True).clone()
            outs.append((eps, [(d["name"], None if d["unm"] is None else d["unm"].clone()) for d in tome.trace], eng.flops, eng.flops_executed))
        tome.reset_global_tokens(); tome.trace = None; tome.draws = None
        res[mode] = outs
    torch.cuda.synchronize()
    for (e0, tr0, f0, x0), (e1, tr1, f1, x1) in zip(res["0"], res["1"]):
        assert torch.isfinite(e1.float()).all()
        assert torch.equal(e0, e1)
        assert [a[0] for a in tr0] == [b[0] for b in tr1]
        for (name, ua), (_, ub) in zip(tr0, tr1):
            if ua is None or ub is None:
                assert ua is None and ub is None, name
            elif ua.shape == ub.shape:
                assert torch.equal(ua, ub), name
            else:                                         # per-sample maps of the de-duplicated block: one row instead of two identical ones
                assert ua.dim() == 2 and torch.equal(ua[0], ua[1]) and torch.equal(ua[0], ub.reshape(-1)), name
        assert f0 == f1 == x0 and x1 < 0.99 * f1           # the reference's FLOPs are counted either way; fewer were executed (2.5 % at this small size)


def test_persistent_attention_panels_stay_clean(setup):
    """The attention panels are zero-initialised ONCE per shape and reused by every chunk, block and pass that meets the shape again; the producers never write
    their padding (unet.py `_attn_panels` / `_q_panel`).  Guard: a pass over other inputs and other chunk lengths in between must leave a later pass
    exactly what it computes on freshly zeroed panels -- if any kernel ever wrote outside its rows / columns the bits would differ."""
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    sd = setup[0]
    Hh, Ww, t = 16, 24, 601.0
    g = torch.Generator(device="cuda").manual_seed(9)
    text = torch.randn(2, 77, 768, device="cuda", generator=g).half()

    def x_of(Fs, scale):
        return (scale * torch.randn(2 * sum(Fs), Hh, Ww, 8, device="cuda", generator=g)).half()
    A, B = ([4, 3, 1], [(1, 0.9), (2, 0.2), (0, 0.6)]), ([4, 4, 2], [(3, 0.3), (0, 0.8), (1, 0.1)])
    xa, xb = x_of(A[0], 3.0), x_of(B[0], 1.0)

    def run(eng, tome, Fs, draws, x):
        tome.reset_global_tokens()
        tome.draws = list(draws)
        return eng.forward_many(x, Fs, Hh, Ww, t, text).clone()
    tome = VidToMe("cuda", seed=5)
    eng = UNetEngine(sd, "cuda", tome)
    run(eng, tome, *A, xa)                                 # fills the panel caches with A's (larger-valued, differently sized) sequences
    n_cached = len(eng.__dict__.get("_panel_cache", {})) + len(eng.__dict__.get("_qpanels", {}))
    assert n_cached > 0
    used = run(eng, tome, *B, xb)                          # ... B on the used panels
    eng.__dict__.pop("_panel_cache", None); eng.__dict__.pop("_qpanels", None)
    fresh = run(eng, tome, *B, xb)                         # ... and on freshly zeroed ones
    torch.cuda.synchronize()
    assert torch.isfinite(fresh.float()).all() and torch.equal(used, fresh)
