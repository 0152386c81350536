"""GPU: the path-1 kernels at BASELINE.json configs[1]'s FULL sizes (30 frames 960x720 -> latent 90x120, chunks of 4 frames), where a
whole-tensor CPU oracle is out of reach: each kernel is checked through properties that do not depend on the size -- sampled rows against
a plain PyTorch fp32 statement of the same rows, row sums of the softmax, the optimality / counting invariants of the bipartite matching,
and whole-tensor comparison against torch fp32 matmul / conv2d on the GPU where that still fits.
Tolerances: f16 in / f32 accumulate / f16 out -> rel-L2 <= 2e-3 per op (3e-3 for attention, whose P is rounded to f16)."""
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


@pytest.mark.parametrize("na,nb,C,ratio", [(32400, 10800, 320, 0.6),     # local (random-frame) merge of a 4-frame chunk at level 0
                                           (23760, 23760, 320, 0.5),     # global merge against an equally long bank
                                           (8100, 2700, 640, 0.6)])      # level 1
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
    assert not ws[: na * 8].any()                       # the key array is left cleared for the next match (documented contract)
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


def test_unet_pass_full_size_deterministic():
    """One block-major UNet pass over ALL chunks of a config-2 step (8 chunks, 30 frames, latent 90x120): finite, and bit-identical when
    repeated from the same bank state and draws (deterministic GroupNorm, fixed K-split rule, order-free matching keys), with the
    matching chain on its side stream or on the main stream."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd import sd15
    from tc_light_amd.unet import UNetEngine
    from tc_light_amd.vidtome import VidToMe
    sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
    eng = UNetEngine(sd, "cuda", VidToMe("cuda", seed=1))
    g = torch.Generator(device="cuda").manual_seed(5)
    text = torch.randn(2, 154, 768, device="cuda", generator=g).half()
    Fs = [2] + [4] * 7
    x = torch.randn(2 * sum(Fs), 90, 120, 8, device="cuda", generator=g).half()
    import os
    outs = []
    for k in range(3):
        if k == 2:
            os.environ["TCL_TOME_STREAM"] = "0"          # matching chain on the main stream: same kernels, same data, no overlap
        eng.tome.reset_global_tokens()
        eng.tome.draws = [(min(1, f - 1) if f > 1 else -1, 0.25 + 0.1 * i) for i, f in enumerate(Fs)]
        outs.append(eng.forward_many(x, Fs, 90, 120, 801.0, text).clone())
    os.environ.pop("TCL_TOME_STREAM", None)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])                 # the side-stream schedule changes timing only
