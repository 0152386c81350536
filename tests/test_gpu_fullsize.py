"""GPU: the path-1 kernels at BASELINE.json configs[1]'s FULL sizes (30 frames 960x720 -> latent 90x120, chunks of 4 frames), where a
whole-tensor CPU oracle is out of reach: each kernel is checked through properties that do not depend on the size -- sampled rows against
a plain PyTorch fp32 statement of the same rows, row sums of the softmax, the optimality / counting invariants of the bipartite matching,
and whole-tensor comparison against torch fp32 matmul / conv2d on the GPU where that still fits.
Tolerances: f16 in / f32 accumulate / f16 out -> rel-L2 <= 2e-3 per op (3e-3 for attention, whose P is rounded to f16)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
H = torch.float16
I32 = torch.int32


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd.lib import lib
    return lib()


def st():
    return torch.cuda.current_stream().cuda_stream


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def ws_bytes(n):
    return torch.empty(int(n), dtype=torch.uint8, device="cuda")


@pytest.mark.parametrize("d,B,Tq,Tk,kv_div", [
    (40, 2, 35640, 35640, 1),      # level-0 self-attention over one merged 4-frame chunk + bank (two-query-block kernel, 4-slot ring)
    (40, 2, 47520, 47520, 1),      # the same at BASELINE config 3 (1280x720: 31 680 local + bank); even count of full tiles + 1 masked
    (40, 2, 16400, 17000, 1),      # two-query-block kernel with an ODD count of full key tiles + a masked one (tail loop takes two tiles)
    (40, 2, 16400, 16448, 1),      # ... and with no padded keys at all (257 full tiles: the tail loop takes the last full tile alone)
    (40, 8, 10800, 154, 4),        # level-0 text cross-attention, context shared by the 4 frames of a chunk
    (80, 2, 8910, 8910, 1),        # level-1 self-attention
    (160, 8, 690, 690, 1)])        # level-2 self-attention (no merging)
def test_attention_full_size_sampled_rows(L, d, B, Tq, Tk, kv_div):
    Hh, C = 8, 8 * d
    g = torch.Generator(device="cuda").manual_seed(Tq + d)
    q = torch.randn(B, Tq, C, device="cuda", generator=g).to(H)
    k = torch.randn(B // kv_div, Tk, C, device="cuda", generator=g).to(H)
    v = torch.randn(B // kv_div, Tk, C, device="cuda", generator=g).to(H)
    o = torch.zeros(B, Tq, C, device="cuda", dtype=H)
    wq, wkv = ws_bytes(L.tcl_attention_q_bytes(B, Hh, Tq, d)), ws_bytes(L.tcl_attention_kv_bytes(B // kv_div, Hh, Tk, d))
    L.tcl_attention_f16(q, C, Tq * C, k, C, Tk * C, v, C, Tk * C, o, C, Tq * C, B, Hh, Tq, Tk, d, d ** -0.5, kv_div, 1, wq, wkv, st())
    rows = torch.cat([torch.tensor([0, 1, 31, 32, 63, 64, 127, 128, 255, 256, Tq - 257, Tq - 2, Tq - 1], device="cuda"),
                      torch.randint(0, Tq, (243,), device="cuda", generator=g)]).clamp_(0, Tq - 1)
    qq = q[:, rows].float().view(B, -1, Hh, d).transpose(1, 2)
    kk = k.float().view(-1, Tk, Hh, d).transpose(1, 2).repeat_interleave(kv_div, 0)
    vv = v.float().view(-1, Tk, Hh, d).transpose(1, 2).repeat_interleave(kv_div, 0)
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, -1, C)
    assert rel(o[:, rows], ref) < 3e-3
    # softmax rows sum to one: with V == 1 every output element is the row sum of P divided by itself
    ones = torch.ones_like(v)
    L.tcl_attention_f16(q, C, Tq * C, k, C, Tk * C, ones, C, Tk * C, o, C, Tq * C, B, Hh, Tq, Tk, d, d ** -0.5, kv_div, 1, wq, wkv, st())
    assert (o.float() - 1).abs().max().item() < 2e-3
    assert torch.isfinite(o).all()


def test_attention_rebase_path_spiked_scores(L):
    """The lazy rebase of the running softmax shift (attn.hip: taken on tile 0 and whenever a row maximum climbs > 2^6 above the shift) is a
    rare, data-dependent branch that random data never takes after tile 0 (cdna_hip_programming.md T13 hazard).  Spike some (query, key)
    pairs so that the maximum jumps at chosen LATE tiles -- in the first and in the second tile of a barrier pair, in both query blocks of a
    wave, and in the masked tail tile -- and compare those rows (and their wave neighbours) with an f32 reference."""
    d, B, Tq, Tk, Hh = 40, 2, 16400, 16950, 8
    C = Hh * d
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randn(B, Tq, C, device="cuda", generator=g).to(H)
    k = torch.randn(B, Tk, C, device="cuda", generator=g).to(H)
    v = torch.randn(B, Tk, C, device="cuda", generator=g).to(H)
    spikes = [(5, 64 * 100 + 3), (37, 64 * 101 + 9), (70, 64 * 200 + 1), (300, 64 * 7 + 60), (8000, Tk - 2), (8001, 64 * 264 + 5), (16399, 64 * 150)]
    for qi, kj in spikes:                                   # q.k * scale * log2(e) ~ 9 * |q|^2 / sqrt(40) * 1.44 >> 2^6 above the running shift
        k[:, kj] = (q[:, qi].float() * 9.0).to(H)
    o = torch.zeros(B, Tq, C, device="cuda", dtype=H)
    wq, wkv = ws_bytes(L.tcl_attention_q_bytes(B, Hh, Tq, d)), ws_bytes(L.tcl_attention_kv_bytes(B, Hh, Tk, d))
    L.tcl_attention_f16(q, C, Tq * C, k, C, Tk * C, v, C, Tk * C, o, C, Tq * C, B, Hh, Tq, Tk, d, d ** -0.5, 1, 1, wq, wkv, st())
    rows = sorted({r for qi, _ in spikes for r in range(max(qi - 33, 0), min(qi + 34, Tq))})
    rows = torch.tensor(rows, device="cuda")
    qq = q[:, rows].float().view(B, -1, Hh, d).transpose(1, 2)
    kk, vv = (t.float().view(B, Tk, Hh, d).transpose(1, 2) for t in (k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, -1, C)
    assert torch.isfinite(o).all()
    assert rel(o[:, rows], ref) < 3e-3
    sp = torch.tensor([qi for qi, _ in spikes], device="cuda")
    pos = torch.searchsorted(rows, sp)
    assert rel(o[:, sp], ref[:, pos]) < 3e-3               # the spiked rows themselves: output ~ v[kj] (one key dominates)


@pytest.mark.parametrize("gain,flagged", [(1.6, False), (3.2, True)])
def test_attention_speculative_softmax_guard_and_fallback(L, gain, flagged):
    """head_dim 40, large launches: the softmax runs without row maxima (attn.hip SPEC) -- the shift sits 4 bits above the running maximum, a
    guard on the packed P registers (some P >= 2) triggers the rebase after the tile's PV, and a P beyond the f16 range flags the block for
    the exact-maximum kernel that follows.  gain 1.6: late keys ~8 bits above everything their row has seen -> guard + rebase, NO block may be
    flagged; gain 3.2: ~20 bits above for the paired query (P overflows inside the tile: the blocks holding those rows must be flagged and
    redone) and at most ~9 bits for every other row (no other block may be flagged).  (test_attention_rebase_path_spiked_scores, gain 9,
    flags every block: such a key is 2^16 above the shift for some row of each.)  Both: all rows of the spiked waves and their neighbours
    against an f32 reference."""
    d, B, Tq, Tk, Hh = 40, 2, 16400, 16950, 8
    C = Hh * d
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn(B, Tq, C, device="cuda", generator=g).to(H)
    k = torch.randn(B, Tk, C, device="cuda", generator=g).to(H)
    v = torch.randn(B, Tk, C, device="cuda", generator=g).to(H)
    spikes = [(9, 64 * 90 + 5), (40, 64 * 91 + 33), (77, 64 * 180 + 63), (301, 64 * 3 + 1), (302, 64 * 4 + 2), (9000, Tk - 1), (9001, 64 * 260), (16399, 64 * 151 + 7)]
    for qi, kj in spikes:
        k[:, kj] = (q[:, qi].float() * gain).to(H)
    o = torch.zeros(B, Tq, C, device="cuda", dtype=H)
    wq, wkv = torch.zeros(L.tcl_attention_q_bytes(B, Hh, Tq, d), dtype=torch.uint8, device="cuda"), ws_bytes(L.tcl_attention_kv_bytes(B, Hh, Tk, d))
    L.tcl_attention_f16(q, C, Tq * C, k, C, Tk * C, v, C, Tk * C, o, C, Tq * C, B, Hh, Tq, Tk, d, d ** -0.5, 1, 1, wq, wkv, st())
    Tqp, nblk = -(-Tq // 256) * 256, B * Hh * -(-Tq // 256)
    off = -(-(B * Hh * Tqp * 48 * 2) // 256) * 256                   # per-block flags behind the Q panel (attn.hip)
    flags = wq[off:off + 4 * nblk].view(torch.int32)
    assert set(flags.unique().tolist()) <= {0, 1}
    nflag = int(flags.sum())
    assert (nflag > 0) == flagged and nflag <= B * Hh * 4, nflag                  # the spiked rows sit in 4 of the 65 query blocks
    rows = torch.tensor(sorted({r for qi, _ in spikes for r in range(max(qi - 65, 0), min(qi + 66, Tq))}), device="cuda")
    qq = q[:, rows].float().view(B, -1, Hh, d).transpose(1, 2)
    kk, vv = (t.float().view(B, Tk, Hh, d).transpose(1, 2) for t in (k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, -1, C)
    assert torch.isfinite(o).all()
    assert rel(o[:, rows], ref) < 3e-3
    sp = torch.tensor([qi for qi, _ in spikes], device="cuda")
    assert rel(o[:, sp], ref[:, torch.searchsorted(rows, sp)]) < 3e-3


@pytest.mark.parametrize("qscale,sink", [(3.0, 0.0), (6.0, 0.0), (1.0, 14.0), (1.0, 18.0)])
def test_attention_heavy_tail_and_sink_precision(L, qscale, sink):
    """The speculative softmax keeps P <= 2^-3, i.e. a weight below 2^-21 of its row's maximum vanishes (2^-24 in an f16 flash kernel with
    P <= 1).  Logit distributions where that band carries mass: heavy tails (q scaled: logit std 4.3 / 8.7 bits) and a sink key 20 / 26 bits
    above the bulk of its row.  Sampled rows against the f32 reference stay inside half of north_star's 1e-3."""
    d, B, Tq, Hh = 40, 2, 16400, 8
    C = Hh * d
    g = torch.Generator(device="cuda").manual_seed(3)
    q = (torch.randn(B, Tq, C, device="cuda", generator=g) * qscale).to(H)
    k = torch.randn(B, Tq, C, device="cuda", generator=g).to(H)
    v = torch.randn(B, Tq, C, device="cuda", generator=g).to(H)
    if sink:
        u = torch.randn(C, device="cuda", generator=g)
        u = (u.view(Hh, d) / u.view(Hh, d).norm(dim=1, keepdim=True)).reshape(C) * (sink * d ** 0.5) ** 0.5
        q = (q.float() + u).to(H)
        k[:, 5000] = u.to(H)
    o = torch.empty_like(q)
    wq, wkv = ws_bytes(L.tcl_attention_q_bytes(B, Hh, Tq, d)), ws_bytes(L.tcl_attention_kv_bytes(B, Hh, Tq, d))
    L.tcl_attention_f16(q, C, Tq * C, k, C, Tq * C, v, C, Tq * C, o, C, Tq * C, B, Hh, Tq, Tq, d, d ** -0.5, 1, 1, wq, wkv, st())
    rows = torch.arange(0, Tq, 37, device="cuda")
    qq = q[:, rows].float().view(B, -1, Hh, d).transpose(1, 2)
    kk, vv = (t.float().view(B, Tq, Hh, d).transpose(1, 2) for t in (k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, -1, C)
    assert torch.isfinite(o).all()
    assert rel(o[:, rows], ref) < 7e-4


def test_attention_trained_like_logits(L):
    """Speculative softmax on logits shaped like a trained UNet's self-attention rather than white noise: every head has its
    own temperature (logit std from 0.5 to 9 bits across the 8 heads), token 0 is an attention SINK that every query scores 12-25 bits above
    the bulk, and keys carry a smooth positional component so neighbouring tokens score alike (long runs of tiles where the running maximum
    does not move, then jumps).  Sampled rows against the f32 reference; the result must also equal the exact-maximum kernel's to f16 rounding."""
    d, B, Tq, Hh = 40, 2, 20000, 8
    C = Hh * d
    g = torch.Generator(device="cuda").manual_seed(11)
    temp = torch.tensor([0.35, 0.6, 1.0, 1.6, 2.5, 4.0, 6.0, 9.0], device="cuda").repeat_interleave(d)        # per-head query scale
    pos = torch.linspace(0, 6.28318 * 3, Tq, device="cuda")[:, None]
    wave = torch.cat([torch.sin(pos * (1 + j)) for j in range(4)], 1) @ torch.randn(4, C, device="cuda", generator=g) * 0.7
    q = ((torch.randn(B, Tq, C, device="cuda", generator=g) + wave) * temp).to(H)
    k = (torch.randn(B, Tq, C, device="cuda", generator=g) + wave).to(H)
    v = torch.randn(B, Tq, C, device="cuda", generator=g).to(H)
    u = torch.randn(C, device="cuda", generator=g)
    u = (u.view(Hh, d) / u.view(Hh, d).norm(dim=1, keepdim=True)).reshape(C)
    k[:, 0] = (u * 9.0).to(H)                                                    # the sink key ...
    q = (q.float() + u * temp.clamp(max=2.0) * 4.0).to(H)                       # ... that every query leans towards
    o = torch.empty_like(q)
    wq, wkv = ws_bytes(L.tcl_attention_q_bytes(B, Hh, Tq, d)), ws_bytes(L.tcl_attention_kv_bytes(B, Hh, Tq, d))
    L.tcl_attention_f16(q, C, Tq * C, k, C, Tq * C, v, C, Tq * C, o, C, Tq * C, B, Hh, Tq, Tq, d, d ** -0.5, 1, 1, wq, wkv, st())
    rows = torch.arange(0, Tq, 41, device="cuda")
    qq = q[:, rows].float().view(B, -1, Hh, d).transpose(1, 2)
    kk, vv = (t.float().view(B, Tq, Hh, d).transpose(1, 2) for t in (k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, -1, C)
    assert torch.isfinite(o).all()
    l2 = lambda t: t.view(B, -1, Hh, d).pow(2).sum(dim=(0, 1, 3)).sqrt()
    per_head = (l2(o[:, rows].float() - ref) / l2(ref)).cpu()
    print("[attention, trained-like logits] rel-L2 per head:", [f"{x:.1e}" for x in per_head.tolist()])
    assert rel(o[:, rows], ref) < 7e-4 and per_head.max() < 1.5e-3


@pytest.mark.parametrize("na,nb,C,ratio", [(32400, 10800, 320, 0.6),     # local (random-frame) merge of a 4-frame chunk at level 0
                                           (23760, 23760, 320, 0.5),     # global merge against an equally long bank
                                           (8100, 2700, 640, 0.6),       # level 1
                                           (43200, 14400, 320, 0.6),     # BASELINE config 3 (1280x720): local merge of a 4-frame chunk, level 0
                                           (31680, 31680, 320, 0.5),     # config 3: global merge against the bank
                                           (32400, 10800, 320, 0.9),     # BASELINE config 4 (tclight_bkgd_robotwin.yaml: local 0.9 / global 0.8)
                                           (14040, 14040, 320, 0.8)])    # config 4: 10800 + 3240 local tokens against an equally long bank
def test_tome_match_full_size_invariants(L, na, nb, C, ratio):
    g = torch.Generator(device="cuda").manual_seed(na)
    T, Bt = na + nb, 2
    base = torch.randn(1, nb, C, device="cuda", generator=g)
    x = torch.cat([base.repeat(1, -(-na // nb), 1)[:, :na], base], 1) + 0.4 * torch.randn(Bt, T, C, device="cuda", generator=g)
    x = x.to(H)
    metric = torch.empty_like(x)
    L.tcl_tome_normalize_f16(x, metric, Bt * T, C, st())
    a_pos = torch.arange(0, na, dtype=I32, device="cuda")
    b_pos = torch.arange(na, T, dtype=I32, device="cuda")
    r = min(na, int(na * ratio))
    ws = torch.zeros(L.tcl_tome_match_workspace_bytes(na), dtype=torch.uint8, device="cuda")
    mrg = torch.full((na - r + nb,), -1, dtype=I32, device="cuda")
    unm = torch.full((T,), -1, dtype=I32, device="cuda")
    L.tcl_tome_match_f16(metric, T * C, Bt, C, a_pos, na, b_pos, nb, r, mrg, unm, ws, st())
    torch.cuda.synchronize()
    assert not ws[:2048].any() and not ws[4096: 4096 + na * 8].any()      # histograms + key array are left cleared for the next match (documented contract)
    nun = na - r
    u = unm[:na].long()
    merged = u >= nun
    # counting invariants (merge.py:90-99: exactly r src tokens are merged, the others keep a slot of their own, dst follow)
    assert int(merged.sum()) == r
    slots = u[~merged]
    assert torch.equal(slots, torch.arange(nun, device="cuda"))              # unmerged src keep index order
    assert torch.equal(mrg[:nun].long(), torch.nonzero(~merged).flatten())
    assert torch.equal(mrg[nun:].long(), torch.arange(na, T, device="cuda"))
    assert torch.equal(unm[na:].long(), nun + torch.arange(nb, device="cuda"))
    # optimality: the partner of a merged token attains the row maximum of the f16 scores over (batch, dst); the r merged tokens are the
    # r best rows.  Reference scores in fp32 from the same f16 metric, rounded to f16 like the reference's half matmul.
    best = torch.full((na,), -2.0, device="cuda")
    chosen = torch.full((na,), -2.0, device="cuda")
    partner = (u - nun).clamp_min(0)
    for b in range(Bt):
        for lo in range(0, na, 8192):
            hi = min(lo + 8192, na)
            s = (metric[b, lo:hi].float() @ metric[b, na:].float().t()).to(H).float()
            best[lo:hi] = torch.maximum(best[lo:hi], s.max(1).values)
            chosen[lo:hi] = torch.maximum(chosen[lo:hi], s.gather(1, partner[lo:hi, None]).flatten())
    ulp = 2.0 ** -10                                    # f16 spacing just below 1: accumulation order may flip one rounding
    gap = (best - chosen)[merged]
    assert (gap <= ulp).all() and (gap == 0).float().mean().item() > 0.98
    assert best[merged].min().item() >= best[~merged].max().item() - ulp


def test_gemm_conv_full_size_vs_torch(L):
    g = torch.Generator(device="cuda").manual_seed(11)
    # level-0 feed-forward of one UNet pass over all chunks: 60 samples x 10800 tokens
    for M, N, K, act in [(648000, 2560, 320, 0), (648000, 320, 1280, 1), (162000, 640, 640, 0)]:
        A = torch.randn(M, K, device="cuda", generator=g).to(H)
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(H)
        b = torch.randn(N, device="cuda", generator=g).to(H)
        Cm = torch.empty(M, N, device="cuda", dtype=H)
        L.tcl_gemm_f16(A, W, b, 0, Cm, M, N, K, K, K, N, N, act, st())
        err2 = ref2 = 0.0
        for lo in range(0, M, 81000):                   # fp32 reference in slabs
            ref = A[lo:lo + 81000].float() @ W.float().t() + b.float()
            if act:
                ref = F.silu(ref)
            err2 += (Cm[lo:lo + 81000].float() - ref).pow(2).sum().item()
            ref2 += ref.pow(2).sum().item()
        assert (err2 / ref2) ** 0.5 < 2e-3
        del A, Cm
    # the UNet's first-level 3x3 convolution over the 60 samples of a pass, and an up-sampling one
    for B, Hh, Ww, Cin, Cout, up in [(60, 90, 120, 320, 320, None), (60, 45, 60, 640, 640, (90, 120))]:
        x = torch.randn(B, Hh, Ww, Cin, device="cuda", generator=g).to(H)
        w = (torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5).to(H)
        b = torch.randn(Cout, device="cuda", generator=g).to(H)
        Ho, Wo = up if up else (Hh, Ww)
        y = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=H)
        L.tcl_conv3x3_f16(x, w.reshape(Cout, 9 * Cin), b, 0, y, B, Hh, Ww, Cin, Cout, 1, 1, up[0] if up else 0, up[1] if up else 0, 0, st())
        err2 = ref2 = 0.0
        for lo in range(0, B, 10):
            xin = x[lo:lo + 10].float().permute(0, 3, 1, 2)
            if up:
                xin = F.interpolate(xin, size=up, mode="nearest")
            ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
            err2 += (y[lo:lo + 10].float() - ref).pow(2).sum().item()
            ref2 += ref.pow(2).sum().item()
        assert (err2 / ref2) ** 0.5 < 2e-3


@pytest.mark.parametrize("Hh,Ww,Fs,Lt", [(90, 120, [2] + [4] * 7, 154),       # config 2: the 8 xy chunks of a 30-frame step
                                         (90, 160, [3] + [4] * 9, 154),       # config 3: one rank's 38-frame block (xy chunks at 1280x720)
                                         (64, 90, [4] * 10, 77)])             # config 3: yt planes of a 64-frame window (64 x 90 "images"), 10 column chunks
def test_unet_pass_full_size_deterministic(Hh, Ww, Fs, Lt):
    """One block-major UNet pass over the chunks of a step at BASELINE sizes: finite, and bit-identical when repeated from the same bank
    state and draws (deterministic GroupNorm, fixed K-split rule, order-free matching keys), with the matching chain on its side stream
    or on the main stream."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd import sd15
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    eng = UNetEngine(sd, "cuda", VidToMe("cuda", seed=1))
    g = torch.Generator(device="cuda").manual_seed(5)
    text = torch.randn(2, Lt, 768, device="cuda", generator=g).half()
    x = torch.randn(2 * sum(Fs), Hh, Ww, 8, device="cuda", generator=g).half()
    import os
    outs = []
    for k in range(3):
        if k == 2:
            os.environ["TCL_TOME_STREAM"] = "0"          # matching chain on the main stream: same kernels, same data, no overlap
        eng.tome.reset_global_tokens()
        eng.tome.draws = [(min(1, f - 1) if f > 1 else -1, 0.25 + 0.1 * i) for i, f in enumerate(Fs)]
        outs.append(eng.forward_many(x, Fs, Hh, Ww, 801.0, text).clone())
    os.environ.pop("TCL_TOME_STREAM", None)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])                 # the side-stream schedule changes timing only


@pytest.mark.parametrize("N,F,randf,T2,src_len,ratio,C", [(14400, 4, 2, 0, 0, 0.6, 320),        # config 3 local merge, dst frame in the middle (two src runs)
                                                         (3600, 4, 2, 0, 0, 0.6, 640),    # the same at level 1 (C = 640: the 160-VGPR strip, round 4)
                                                         (0, 0, 0, 15840, 7920, 0.5, 640),     # level-1 two-set (global) merge
                                                         (1440, 3, 0, 0, 0, 0.9, 640),    # level-1 yt plane, ragged tiles
                                                         (14400, 4, 0, 0, 0, 0.6, 320),        # dst = first frame
                                                         (5760, 3, 2, 0, 0, 0.9, 320),         # yt plane of a 64-frame window, dst = last frame, ragged tiles
                                                         (0, 0, 0, 63360, 31680, 0.5, 320),    # config 3 two-set (global) merge, local tokens are src
                                                         (0, 0, 0, 40777, 17017, 0.8, 320),    # ragged sizes, config 4's global ratio
                                                         (300, 4, 1, 0, 0, 0.6, 320),          # smaller than one dst split; last dst tile moved back
                                                         (100, 4, 1, 0, 0, 0.6, 320)])         # fewer than 128 dst tokens: the strip kernel hands over to the tile kernel
def test_tome_match_strip_kernel_equals_tile_kernel(L, N, F, randf, T2, src_len, ratio, C):
    """k_tome_match320 (src strip in registers, running maximum over the dst sweep; taken through tcl_tome_match_affine_f16 for C = 320 and,
    since round 4, C = 640) must produce the SAME maps as the general tile-epilogue kernel (tcl_tome_match_f16), bit for bit: same MFMA, same K
    order, same key.  The workspace must come back all-zero except the two result words of the control block."""
    g = torch.Generator(device="cuda").manual_seed(N + T2 + randf)
    if N:
        T = F * N
        idx = torch.arange(T, dtype=I32, device="cuda")
        dst = (idx // N) % F == randf
        a_pos, b_pos = idx[~dst].contiguous(), idx[dst].contiguous()
        aff = (randf * N, N, randf * N)
    else:
        T = T2
        a_pos, b_pos = torch.arange(0, src_len, dtype=I32, device="cuda"), torch.arange(src_len, T, dtype=I32, device="cuda")
        aff = (src_len, 0, src_len)
    na, nb = a_pos.numel(), b_pos.numel()
    x = torch.randn(2, T, C, device="cuda", generator=g)
    x[:, T // 3] = x[:, T // 2]                          # exact duplicates -> exact score ties: exercises the lowest-index rule
    x[:, 5] = x[:, T - 7]
    x = x.to(H)
    metric = torch.empty_like(x)
    L.tcl_tome_normalize_f16(x, metric, 2 * T, C, st())
    r = min(na, int(na * ratio))
    outs = []
    L.tcl_tome_strip640(1 if C == 640 else 0)             # (the C = 640 strip kernel is opt-in: slower beside the flash kernel)
    for affine in (False, True):
        ws = torch.zeros(L.tcl_tome_match_workspace_bytes(na), dtype=torch.uint8, device="cuda")
        mrg = torch.full((na - r + nb,), -1, dtype=I32, device="cuda")
        unm = torch.full((T,), -1, dtype=I32, device="cuda")
        if affine:
            L.tcl_tome_match_affine_f16(metric, T * C, 2, C, a_pos, na, b_pos, nb, r, aff[0], aff[1], aff[2], mrg, unm, ws, st())
        else:
            L.tcl_tome_match_f16(metric, T * C, 2, C, a_pos, na, b_pos, nb, r, mrg, unm, ws, st())
        torch.cuda.synchronize()
        assert not ws[:3072].any() and not ws[3072 + 16:].any()       # (ints 768..771 of the control block: tickets (zero again), thr, take)
        outs.append((mrg, unm))
    L.tcl_tome_strip640(0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_config3_windows_and_shards():
    """BASELINE config 3 host logic at size: 300 frames, window 64 -> 5 windows starting [0, 59, 118, 177, 236] with overlaps [5, 5, 5, 5]
    (generate.py:246-260), frames sharded 38/38/38/38/37/37/37/37, and the per-rank deal of the 5 x 40 yt items covers every
    (window, column) exactly once."""
    from tc_light_amd import hostlogic as HL
    from tc_light_amd.parallel import Dist
    starts, ovl = HL.temporal_windows(300, 64)
    assert starts == [0, 59, 118, 177, 236] and ovl == [5, 5, 5, 5]
    assert [HL.shard_range(300, r, 8)[1] - HL.shard_range(300, r, 8)[0] for r in range(8)] == [38] * 4 + [37] * 4
    items = [(sl, k) for sl in starts for k in range(40)]
    seen = sorted(it for r in range(8) for it in Dist(r, 8).my_items(items))
    assert seen == sorted(items)
    assert max(len(Dist(r, 8).my_items(items)) for r in range(8)) == 25


def _gpu_tracks(n, h, w, reuse, seed):
    """synth.track_ids on the device (the numpy version takes a minute at 300 x 1280 x 720): frame k re-uses frame k-1's ids shifted by one
    pixel with probability `reuse`, fresh ids otherwise; ids are unique within a frame like get_flowid's."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    ids = torch.empty(n, h, w, dtype=torch.int64, device="cuda")
    ids[0] = torch.arange(h * w, device="cuda").view(h, w)
    last = h * w
    for k in range(1, n):
        fresh = torch.rand(h, w, device="cuda", generator=g) > reuse
        fresh[:, 0] = True
        cur = torch.roll(ids[k - 1], 1, dims=1)
        cnt = int(fresh.sum())
        cur[fresh] = last + torch.arange(cnt, device="cuda")
        last += cnt
        ids[k] = cur
    return ids.reshape(-1).to(torch.int32), last


def test_config5_stage2_full_size_properties():
    """BASELINE config 5 (stage 2 only, 300 x 1280 x 720, K >= 1e8 codebook rows) through size-independent properties of
    tcl_unique_tensor_opt / tcl_unique_tensor_grad (the oracle would need hours here):
      * zero iterations: the final gather of the scatter-mean initialised codebook reproduces, per pixel, the mean of its track;
      * one iteration moves exactly the rows of the mini-batch's frames (cur and prev) -- Adam's first step is +-lr where the gradient is
        non-zero and 0 elsewhere -- and every moved row moves by at most lr (SH units);
      * the iteration-level C entry point + tcl_adam_step reproduces the whole-stage driver (same kernels, 1e-6);
      * three iterations keep everything finite and lower the loss on a repeated batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd import post_opt as P
    from tc_light_amd.lib import lib
    L = lib()
    n, h, w = 300, 720, 1280
    inv, k = _gpu_tracks(n, h, w, reuse=0.6, seed=3)
    assert k >= 100_000_000 and k < 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(1)
    base = torch.rand(1, 3, h + 8, w + 300, device="cuda", generator=g)
    base = torch.nn.functional.avg_pool2d(base, 9, stride=1, padding=4)
    ed = torch.stack([base[0, :, 4:4 + h, i:i + w] for i in range(n)]).contiguous()        # frame i = the base shifted by i px (matches the ids' shift)
    ed = (ed * (1 + 0.03 * torch.randn(n, 3, 1, 1, device="cuda", generator=g)) + 0.01 * torch.randn(ed.shape, device="cuda", generator=g)).clamp_(0, 1)
    flows = torch.zeros(n, 2, h, w, device="cuda"); flows[1:, 0] = -1.0        # frame i (x) = frame i-1 (x - 1)
    masks = torch.ones(n, 1, h, w, device="cuda")
    ds = P.OptDataset(ed, flows, masks, device="cuda")
    # ---- zero iterations
    out0, feat0, _ = P.unique_tensor_optimization(ds, inv, np.zeros((0, 16), np.int32), batch_size=16, k=k)
    feat0 = feat0.t().contiguous().clone()                                     # planar [3,K]
    cnt = torch.bincount(inv.long(), minlength=k).float()
    for c in range(3):
        mean = torch.zeros(k, device="cuda").index_add_(0, inv.long(), ed[:, c].reshape(-1)) / cnt.clamp_min(1)
        assert (out0[:, c].reshape(-1) - mean[inv.long()].clamp(0, 1)).abs().max().item() < 2e-5
    del out0
    # ---- one iteration
    batch = np.array([[17, 250, 3, 120, 299, 64, 180, 1, 33, 90, 210, 5, 270, 150, 44, 0]], np.int32)
    _, feat1, l1 = P.unique_tensor_optimization(ds, inv, batch, batch_size=16, k=k)
    feat1 = feat1.t().contiguous()
    lr = 0.05 * 16 / n
    moved = ((feat1 - feat0).abs() > 0.25 * lr).any(0)          # (the scatter-mean initialisation sums with float atomics: two runs differ by ~1e-7)
    frames_touched = sorted({int(f) for f in batch[0]} | {max(int(f) - 1, 0) for f in batch[0]})
    touched = torch.zeros(k, dtype=torch.bool, device="cuda")
    for f in frames_touched:
        touched[inv[f * h * w:(f + 1) * h * w].long()] = True
    assert not (moved & ~touched).any()                                        # rows outside the batch's frames did not move
    assert moved.float().sum().item() > 0.5 * touched.float().sum().item()
    assert (feat1 - feat0).abs().max().item() <= lr * 1.001
    assert torch.isfinite(l1).all()
    # ---- iteration-level API == whole-stage driver
    from tc_light_amd.lib import stream
    feat = feat0.clone()
    gbuf, m, v = (torch.zeros_like(feat) for _ in range(3))
    cat = torch.from_numpy(np.concatenate([batch[0], np.maximum(batch[0] - 1, 0)]).astype(np.int32)).cuda()
    ws = torch.empty(L.tcl_stage_workspace_bytes(16, h, w), dtype=torch.uint8, device="cuda")
    lp = torch.zeros(1, device="cuda")
    L.tcl_unique_tensor_grad(ds.edited_images, flows, masks, ds.flow_shift, inv, n, h, w, k, 1, cat, 16, 16, 15, 0.2, 0.8, 0.05, feat, gbuf, lp, ws, stream())
    L.tcl_adam_step(feat, gbuf, m, v, 3 * k, lr, 0.9, 0.999, 1e-15, 1, stream())
    assert abs(float(lp) - float(l1[0])) < 2e-5 * abs(float(l1[0]))
    d = (feat - feat1).abs()
    assert (d > 1e-6).float().mean().item() < 1e-3                             # float-atomic order: a few rounding-noise rows take +-lr
    del feat, gbuf, m, v, feat1
    # ---- three iterations on one batch: finite, decreasing
    _, f3, l3 = P.unique_tensor_optimization(ds, inv, np.repeat(batch, 3, 0), batch_size=16, k=k)
    assert torch.isfinite(l3).all() and torch.isfinite(f3).all() and float(l3[2]) < float(l3[0])


def test_config4_background_blend_full_size():
    """BASELINE config 4 geometry: prepare_data's matte + blend at 960x720 (the RMBG input is 1152x896 through the reference's transposed
    resize, generate.py:151-153) against the oracle's matte on 2 of the frames; local 0.9 / global 0.8 matching is covered by
    test_tome_match_full_size_invariants."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from types import SimpleNamespace
    from oracle import rmbg as OR
    from tc_light_amd.generate import Generator
    from tc_light_amd.rmbg import RMBGEngine, random_state_dict
    sd = random_state_dict(3)
    eng = RMBGEngine(sd, "cuda")
    stub = SimpleNamespace(dev=torch.device("cuda"), tome=SimpleNamespace(args=dict(target_stride=4)))
    gen = Generator(stub, None, dict(noise_mode="same", local_merge_ratio=0.9, global_merge_ratio=0.8), rmbg=eng)
    assert gen.unet.tome.args["local_merge_ratio"] == 0.9 and gen.unet.tome.args["global_merge_ratio"] == 0.8
    import synth
    fr = synth.video_clip(2, 720, 960, seed=9)["frames"]
    bg = torch.from_numpy(np.random.default_rng(1).random((1, 3, 720, 960)).astype(np.float32))
    gen.prepare_data(fr.cuda(), background=bg.cuda())
    with torch.no_grad():
        alpha = OR.estimate_alpha(sd, fr)
    err = (gen.frames.cpu() - (alpha * fr + (1 - alpha) * bg)).abs()
    assert err.max().item() < 2e-3 and err.mean().item() < 5e-5, (err.max().item(), err.mean().item())
    assert gen.init_noise.shape == (2, 4, 90, 120)


def test_kernels_beyond_2g_elements_and_4gib_operands(L):
    """Block-major passes larger than the default 1.5 M level-0 tokens carry activations with more than 2^31 elements / 4 GiB (3.46 M rows x 1280
    channels at 120 frames of 1280x720 in ONE pass): every kernel family of the UNet pass is run once on such a tensor and checked on sampled rows
    around the 2^31-element and 4-GiB marks and at the end against plain torch fp32 of those rows.  (The 8-phase GEMM addresses A relative to the
    block's first row / image, so its 32-bit DMA offsets only span one tile.)"""
    g = torch.Generator(device="cuda").manual_seed(5)
    M = 3_460_000
    rows = torch.tensor([0, 1, 255, 256, 1_677_721, 1_677_722, 1_677_800, 3_355_443, 3_355_444, 3_400_001, M - 257, M - 2, M - 1], device="cuda")
    # ---- dense GEMM, K = 1280 (A = 8.9 GB): 8-phase 256x320 / LDS-DMA tile / automatic choice; residual + bias
    A = torch.empty(M, 1280, device="cuda", dtype=H)
    for i in range(0, M, 432_500):
        A[i:i + 432_500] = torch.randn(min(432_500, M - i), 1280, device="cuda", generator=g).to(H)
    W = (torch.randn(320, 1280, device="cuda", generator=g) / 1280 ** 0.5).to(H)
    b = torch.randn(320, device="cuda", generator=g).to(H)
    R = torch.randn(M, 320, device="cuda", generator=g).to(H)
    ref = (A[rows].float() @ W.float().t() + b.float()).to(H).float() + R[rows].float()
    outs = []
    try:
        for cfg in (14, 1, 0):
            L.tcl_gemm_tune(cfg, 1 if cfg else 0)
            C = torch.empty(M, 320, device="cuda", dtype=H)
            L.tcl_gemm_f16(A, W, b, R, C, M, 320, 1280, 1280, 1280, 320, 320, 0, st())
            assert rel(C[rows], ref) < 2e-3, cfg
            outs.append(C)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        del outs, C, R
        # ---- GEGLU feed-forward shape: K = 320 -> N = 2560 (output 3.46 M x 1280 = 4.4e9 elements), strip kernel / 8-phase 256x256
        X = A[:, :320].contiguous()
        W2 = (torch.randn(2560, 320, device="cuda", generator=g) / 320 ** 0.5).to(H)
        b2 = torch.randn(2560, device="cuda", generator=g).to(H)
        f = X[rows].float() @ W2.float().t() + b2.float()
        fr = f.to(H).float()
        ref2 = torch.cat([fr[:, 64 * k:64 * k + 32] * F.gelu(fr[:, 64 * k + 32:64 * k + 64]) for k in range(40)], 1)
        for cfg in (13, 12):
            L.tcl_gemm_tune(cfg, 1)
            C = torch.empty(M, 1280, device="cuda", dtype=H)
            L.tcl_gemm_f16(X, W2, b2, 0, C, M, 2560, 320, 320, 320, 1280, 2560, 2, st())
            assert rel(C[rows], ref2) < 2e-3, cfg
        L.tcl_gemm_tune(0, 0)
        # ---- LayerNorm, GroupNorm (two sources), concat, GEGLU-free elementwise on C (4.4e9 elements)
        ga, be = torch.randn(1280, device="cuda", generator=g).to(H), torch.randn(1280, device="cuda", generator=g).to(H)
        Y = torch.empty_like(C)
        L.tcl_layernorm_f16(C, ga, be, Y, M, 1280, 1e-5, st())
        assert rel(Y[rows], F.layer_norm(C[rows].float(), (1280,), ga.float(), be.float(), 1e-5)) < 2e-3
        Bn, HW = 346, 10000                                                     # 346 samples x 10 000 pixels = M rows
        ws = torch.zeros(int(L.tcl_groupnorm_workspace_bytes(Bn, 1280)), dtype=torch.uint8, device="cuda")
        L.tcl_groupnorm_f16(C, 1280, 0, 0, ga, be, Y, Bn, HW, 32, 1e-5, 1, ws, st())
        for smp in (0, 167, 168, 345):                                          # sample 168 starts at element 2.15e9
            xs = C[smp * HW:(smp + 1) * HW].float()
            want = F.silu(F.group_norm(xs.t()[None], 32, ga.float(), be.float(), 1e-5))[0].t()
            assert rel(Y[smp * HW:(smp + 1) * HW], want) < 2e-3, smp
        del Y
        # ---- implicit 3x3 convolution: 240 images 90x160, 640 -> 320 channels (input 4.4 GB, 2.2e9 elements): 8-phase 256x320 vs the LDS-DMA tile
        del C, X, A
        Bc, Hh, Ww = 240, 90, 160
        x = torch.empty(Bc, Hh, Ww, 640, device="cuda", dtype=H)
        for i in range(0, Bc, 40):
            x[i:i + 40] = torch.randn(40, Hh, Ww, 640, device="cuda", generator=g).to(H)
        w = (torch.randn(320, 9 * 640, device="cuda", generator=g) / (9 * 640) ** 0.5).to(H)
        ys = []
        for cfg in (14, 1):
            L.tcl_gemm_tune(cfg, 1)
            y = torch.empty(Bc, Hh, Ww, 320, device="cuda", dtype=H)
            L.tcl_conv3x3_f16(x, w, b, 0, y, Bc, Hh, Ww, 640, 320, 1, 1, 0, 0, 0, st())
            ys.append(y)
        assert torch.equal(ys[0], ys[1])
        for smp in (0, 116, 117, 239):                                          # image 117 starts at element 1.08e9 x 2 B = beyond 2 GiB; 233+ beyond 4 GiB
            want = F.conv2d(x[smp:smp + 1].permute(0, 3, 1, 2).float(), w.view(320, 3, 3, 640).permute(0, 3, 1, 2).float(), b.float(), padding=1).permute(0, 2, 3, 1)
            assert rel(ys[0][smp:smp + 1], want) < 2e-3, smp
    finally:
        L.tcl_gemm_tune(0, 0)
    torch.cuda.synchronize()


def test_unet_pass_group_size_invariance_beyond_2g_elements():
    """One xy step of 128 frames at 1280x720 latents (90 x 160) through `Generator._unet_xy` with the default 1.5 M-token passes (3 groups) and as ONE
    block-major pass (3.7 M level-0 rows: the 1280-channel feed-forward tensors hold 4.7e9 elements, 9.4 GB).  Every kernel is batch-row
    independent (GroupNorm partitions by HW, the K-split rule is 1 above 16 384 rows, attention runs per chunk, the bank chain visits the chunks
    in the same order), so the noise prediction must be the SAME BITS -- which also checks every kernel of the pass beyond 2^31 elements."""
    from types import SimpleNamespace
    from tc_light_amd import sd15
    from tc_light_amd.generate import Generator
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    dev = torch.device("cuda")
    sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    n, h, w = 128, 90, 160
    g = np.random.default_rng(3)
    x = torch.from_numpy(g.standard_normal((n, 4, h, w)).astype(np.float32)).to(dev).half()
    cc = torch.from_numpy(g.standard_normal((n, 4, h, w)).astype(np.float32)).to(dev).half()
    text = torch.from_numpy(g.standard_normal((2, 154, 768)).astype(np.float32)).to(dev).half()
    chunks = [list(range(i, min(i + 4, n))) for i in range(0, n, 4)]
    outs = []
    for cap in (1_500_000, 16_000_000):
        unet = UNetEngine(sd, dev, VidToMe(dev, seed=7))
        gen = Generator(unet, SimpleNamespace(), dict(max_tokens_per_pass=cap, seed=1))
        gen.h, gen.w = h, w
        noises = torch.zeros_like(x)
        gen._unet_xy(x, cc, chunks, text, 801.0, noises)
        torch.cuda.synchronize()
        assert torch.isfinite(noises.float()).all()
        outs.append(noises)
        del unet, gen
        torch.cuda.empty_cache()
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    assert torch.equal(outs[0], outs[1]), f"one 3.7 M-row pass differs from three 1.5 M-token passes: max |d| = {d:.3e}"
