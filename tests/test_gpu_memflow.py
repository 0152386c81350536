"""GPU: the on-demand correlation lookup (csrc/flow.hip through tc_light_amd.memflow.CorrBlock) against the outputs of the reference's
CorrBlock (tests/golden/memflow_corr.npz) and against the CPU oracle on a larger seeded case.  f32 with a different summation order:
3e-5 abs on values of order 1."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda"


def test_corr_lookup_vs_reference_golden(dev):
    from tc_light_amd.memflow import CorrBlock
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "memflow_corr.npz"))
    for tag in ("a", "b"):
        f1, f2, co, ref = (torch.from_numpy(G[f"{tag}_{k}"]) for k in ("f1", "f2", "coords", "out"))
        got = CorrBlock(f1.to(dev), f2.to(dev), num_levels=4, radius=4)(co.to(dev)).cpu()
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() < 3e-5


def test_corr_lookup_vs_oracle_full_size(dev):
    """1280x720 / 8 feature maps (the reference's volume would be 829 MB per pair): property + oracle on a row subset."""
    from oracle import memflow as OM
    from tc_light_amd.memflow import CorrBlock
    g = torch.Generator().manual_seed(3)
    B, D, H, W = 1, 256, 90, 160
    f1, f2 = torch.randn(B, D, H, W, generator=g), torch.randn(B, D, H, W, generator=g)
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    co = torch.stack([xs, ys])[None] + 4 * torch.randn(B, 2, H, W, generator=g)
    got = CorrBlock(f1.to(dev), f2.to(dev))(co.to(dev)).cpu()
    ref = OM.corr_lookup(f1, f2, co)
    assert (got - ref).abs().max().item() < 5e-5
    # identity flow, level 0, window centre (a = b = r): <f1, f2> at the same pixel
    c0 = torch.stack([xs, ys])[None]
    cen = CorrBlock(f1.to(dev), f2.to(dev))(c0.to(dev))[:, 4 * 9 + 4].cpu()
    assert (cen - (f1 * f2).sum(1) / 16.0).abs().max().item() < 5e-5


def test_encoders_vs_reference_golden(dev):
    """EncoderEngine (f16 NHWC, f32 accumulate) against the reference BasicEncoder outputs (f32) on the seeded weights: ~20 chained
    convolutions with f16 activations -> rel-L2 5e-3."""
    from tc_light_amd import memflow as MF
    N = np.load(os.path.join(os.path.dirname(__file__), "golden", "memflow_net.npz"))
    img = torch.from_numpy(N["img"]).to(dev)
    for name, norm in (("fnet", "instance"), ("cnet", "batch")):
        sd = MF.seeded_state_dict(MF.encoder_param_shapes("", norm), int(N[name + "_seed"]))
        eng = MF.EncoderEngine(sd, "", norm, dev)
        f, (h, w) = eng.forward(img)
        got = f.view(2, h, w, 256).permute(0, 3, 1, 2).float().cpu()
        ref = torch.from_numpy(N[name])
        assert got.shape == ref.shape
        assert ((got - ref).norm() / ref.norm()).item() < 5e-3, name


def test_memflow_engine_vs_reference_golden(dev):
    """MemFlowEngine (f16 activations, f32 accumulate / coordinates) against the reference MemFlowNet + InferenceCore (f32) on the seeded
    weights: three frame pairs, working memory, warm start.  15 GRU iterations with f16 activations: rel-L2 <= 1.5e-2 on the flow
    (measured 1e-3 .. 5e-3, printed)."""
    from tc_light_amd import memflow as MF
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "memflow_full.npz"))
    sd = MF.seeded_state_dict(MF.memflow_param_shapes(), int(G["seed"]))
    eng = MF.MemFlowEngine(sd, dev)
    frames = torch.from_numpy(G["frames"]).to(dev)
    errs = []
    for i in range(3):
        init = None if i == 0 else torch.from_numpy(G[f"init{i}"]).to(dev)
        low, up = eng.step(torch.stack([frames[i], frames[i + 1]])[None], end=(i == 2), flow_init=init)
        for got, key in ((low, f"low{i}"), (up, f"up{i}")):
            ref = torch.from_numpy(G[key])
            assert got.shape == ref.shape
            errs.append(((got.cpu() - ref).norm() / ref.norm()).item())
    print("memflow engine rel-L2 (low0, up0, low1, up1, low2, up2):", errs)
    assert max(errs) < 1.5e-2


def test_estimate_flows_vs_oracle_driver(dev):
    """estimate_flows (video_dataparser.py:63-110,141-156: interleaved future / past pairs on ONE inference core, per-direction warm start,
    zero flow at the sequence ends).  (1) the same call sequence issued by hand on a second engine gives bit-identical flows (the driver adds
    nothing but order and padding); (2) along that sequence every engine step agrees with the CPU oracle stepped in lock-step with the SAME
    warm-start fields (the nearest-neighbour warm start is discontinuous, so each side's own chain would amplify last-bit differences)."""
    from oracle import memflow as OM
    from tc_light_amd import memflow as MF
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "memflow_full.npz"))
    sd = MF.seeded_state_dict(MF.memflow_param_shapes(), int(G["seed"]))
    frames01 = (torch.from_numpy(G["frames"])[:3] + 1) / 2                      # [0,1] frames, as the data parser holds them
    fut, past = MF.estimate_flows(MF.MemFlowEngine(sd, dev), frames01.to(dev))
    assert fut.shape == past.shape == (3, 2, 128, 192)
    assert fut[2].abs().max().item() == 0 and past[0].abs().max().item() == 0
    eng, core, prev = MF.MemFlowEngine(sd, dev), OM.InferenceCore(sd), {True: None, False: None}
    gts = frames01 * 2 - 1
    with torch.no_grad():
        for idx in range(3):
            for is_future in (True, False):
                if idx == (2 if is_future else 0):
                    continue
                tgt = gts[idx + 1] if is_future else gts[idx - 1]
                pair = torch.stack([gts[idx], tgt])[None]
                low, up = eng.step(pair.to(dev), flow_init=prev[is_future])
                assert torch.equal(up[0], (fut if is_future else past)[idx]), (idx, is_future)
                low_o, up_o = core.step(pair, flow_init=None if prev[is_future] is None else prev[is_future].cpu())
                assert ((up.cpu() - up_o).norm() / up_o.norm()).item() < 1.5e-2, (idx, is_future)
                prev[is_future] = MF.forward_interpolate(low[0])[None].to(dev)


def test_step_graph_replay_equals_eager(dev, monkeypatch):
    """Round 6: with TCL_MEMFLOW_GRAPH=1 MemFlowEngine.step replays a captured HIP graph from the third step of a shape on (first eager, second captured).  Ten frame pairs through one
    working memory with warm starts, graphs on against TCL_MEMFLOW_GRAPH=0: every flow bit-identical, and the graph path really replayed (>= 6 replays)."""
    from tc_light_amd import memflow as MF
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "memflow_full.npz"))
    sd = MF.seeded_state_dict(MF.memflow_param_shapes(), int(G["seed"]))
    frames = torch.from_numpy(G["frames"]).to(dev)
    n = frames.shape[0]

    def run(graph):
        monkeypatch.setenv("TCL_MEMFLOW_GRAPH", "1" if graph else "0")             # (opt-in: measured no faster, profiles/r6_ab_memflow_graph_nw.txt)
        eng, outs, init = MF.MemFlowEngine(sd, dev), [], None
        for i in range(10):
            a, b = frames[i % n], frames[(i * 3 + 1) % n]
            low, up = eng.step(torch.stack([a, b])[None], flow_init=init)
            outs.append((low.clone(), up.clone()))
            init = (0.5 * low).contiguous()                                    # a device-side warm start (the driver's scipy one is host work)
        torch.cuda.synchronize()
        return outs, eng
    o1, e1 = run(True)
    o0, _ = run(False)
    for i, ((l1, u1), (l0, u0)) in enumerate(zip(o1, o0)):
        assert torch.equal(l1, l0) and torch.equal(u1, u0), i
    replayed = [v for v in e1._graphs.values() if isinstance(v, dict)]
    assert len(replayed) >= 1 and len(e1._graphs) <= 3, e1._graphs.keys()


def test_corr_lookup_tiled_rows_vs_per_pixel_kernel(dev, monkeypatch):
    """Round 6: lookup_rows through the tile-sharing kernel (f16 feature maps, 8 x 8 pixel tiles share one bounding box of neighbour rows in LDS) against the
    per-pixel f32 kernel on the SAME f16-representable features (the encoder's outputs are f16): (a) a smooth flow -- every tile compact, no fallback; (b) the
    same plus a few torn tiles and windows hanging over every image border -- those tiles take the per-pixel route inside the same call; a 2.6x zoom whose tile
    boxes need two staging bands; (c) a flow that tears
    every tile apart -- all fallback, bit-identical to the per-pixel kernel.  H, W no multiples of 8 (partial tiles).  Levels 1-3 are f16-rounded in the tiled
    kernel: tolerance 2e-3 of the correlation scale."""
    from tc_light_amd.memflow import CorrBlock
    g = torch.Generator().manual_seed(5)
    B, D, H, W = 1, 256, 67, 203                               # (large enough that a box over the whole level-0 map needs more than CT_MAXB bands)
    f1 = torch.randn(B, D, H, W, generator=g).half().float().to(dev)
    f2 = torch.randn(B, D, H, W, generator=g).half().float().to(dev)
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    base = torch.stack([xs, ys])[None]
    smooth = base + torch.stack([2.3 + 1.5 * torch.sin(xs / 30 + ys / 50), -1.2 + 1.0 * torch.cos(ys / 25)])[None]
    torn = smooth.clone()
    torn[:, :, 8:16, 16:24] += 40 * torch.randn(1, 2, 8, 8, generator=g)               # one torn tile
    torn[:, 0, :, :3] -= 7.5; torn[:, 0, :, -3:] += 9.25; torn[:, 1, :2] -= 6.0; torn[:, 1, -2:] += 8.5        # windows over the borders
    wild = base + 60 * torch.randn(1, 2, H, W, generator=g)
    zoom = (base - torch.tensor([W / 2, H / 2]).view(1, 2, 1, 1)) * 2.6 + torch.tensor([W / 2, H / 2]).view(1, 2, 1, 1)     # a 2.6x zoom: a tile's windows span ~31 x 31 points -> 2 bands
    cb = CorrBlock(f1, f2)
    n_tiles = cb._flags.numel()
    for name, co, want_fb in (("smooth", smooth, 0), ("zoom (banded)", zoom, 0), ("torn", torn, None), ("wild", wild, "most")):
        co = co.to(dev).contiguous()
        rows_t = torch.zeros(H * W, 384, dtype=torch.float16, device=dev)
        rows_p = torch.zeros_like(rows_t)
        monkeypatch.setenv("TCL_CORR_TILED", "1")
        cb.lookup_rows(co, rows_t)
        fb = int(cb._flags.sum().item())
        monkeypatch.setenv("TCL_CORR_TILED", "0")
        cb.lookup_rows(co, rows_p)
        torch.cuda.synchronize()
        a, b = rows_t.float().cpu(), rows_p.float().cpu()
        assert torch.equal(a[:, 324:], b[:, 324:])                                     # padding channels untouched
        err = (a - b).abs().max().item()
        print(f"[corr tiled, {name}] fallback tiles {fb} / {n_tiles} (4 levels), max |diff| {err:.2e} of scale {b.abs().max().item():.2f}")
        assert err < 2e-3 * max(1.0, b.abs().max().item()), (name, err)
        if want_fb == 0:
            assert fb == 0
        elif want_fb == "most":
            assert fb >= n_tiles // 4                                                  # at least every level-0 tile (the coarser levels fit the bands)
            lv0 = cb._flags[:n_tiles // 4].bool().cpu()                                 # level 0: every pixel of a flagged tile comes from the f32 kernel -> same bits
            tx = (W + 7) // 8
            tile_of = ((ys.long() // 8) * tx + xs.long() // 8).reshape(-1)
            m = lv0[tile_of]
            assert m.any() and torch.equal(a[m][:, :81], b[m][:, :81])
        else:
            assert 0 < fb < n_tiles // 2


def test_attention_splitkv_equals_single_pass(dev):
    """Round 6: the memory-read attention with the keys cut into chunks that run as batch entries + a merge by the chunks' softmax denominators
    (tcl_attention_splitkv_f16) against the one-pass kernel and against f32 SDPA: 2 / 3 / 5 chunks, the MemFlowNet shapes (one head, head_dim 128, P queries,
    T = P or 2 P keys), logits with a heavy tail so the chunks' maxima differ by many bits."""
    from tc_light_amd.lib import lib, stream
    L = lib()
    g = torch.Generator(device="cuda").manual_seed(4)
    for P_, T in ((1000, 1920), (2250, 5760)):
        qk = torch.randn(P_, 256, device=dev, generator=g).half()
        k = (torch.randn(T, 128, device=dev, generator=g) * torch.linspace(0.2, 3.0, T, device=dev)[:, None]).half()      # later keys score higher: chunk maxima differ
        v = torch.randn(T, 128, device=dev, generator=g).half()
        scale = 128 ** -0.5 * 1.3
        ref = torch.nn.functional.scaled_dot_product_attention(qk[None, None, :, :128].float(), k[None, None].float(), v[None, None].float(), scale=scale)[0, 0]
        one = torch.empty(P_, 128, dtype=torch.float16, device=dev)
        wq = torch.empty(L.tcl_attention_q_bytes(1, 1, P_, 128), dtype=torch.uint8, device=dev)
        wkv = torch.empty(L.tcl_attention_kv_bytes(1, 1, T, 128), dtype=torch.uint8, device=dev)
        L.tcl_attention_f16(qk, 256, P_ * 256, k, 128, T * 128, v, 128, T * 128, one, 128, P_ * 128, 1, 1, P_, T, 128, scale, 1, 1, wq, wkv, stream())
        e_one = ((one.float() - ref).norm() / ref.norm()).item()
        for ns in (2, 3, 5):
            if T % (64 * ns):
                continue
            ws = torch.empty(L.tcl_attention_splitkv_workspace_bytes(ns, 1, P_, T, 128), dtype=torch.uint8, device=dev)
            out = torch.full((P_, 128), float("nan"), dtype=torch.float16, device=dev)
            L.tcl_attention_splitkv_f16(qk, 256, k, 128, v, 128, out, 128, 1, P_, T, 128, scale, ns, ws, stream())
            torch.cuda.synchronize()
            e = ((out.float() - ref).norm() / ref.norm()).item()
            print(f"[split-KV] P {P_} T {T} chunks {ns}: rel-L2 vs f32 SDPA {e:.2e} (one pass {e_one:.2e})")
            assert torch.isfinite(out.float()).all() and e < max(2 * e_one, 2e-3), (ns, e, e_one)
