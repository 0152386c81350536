"""CPU: path-1 oracles (VidToMe, host loop) against the goldens produced by the reference's own code."""
import numpy as np
import torch

from oracle import pipeline as OP
from oracle import vidtome as OV


def test_vidtome_chain(golden):
    g = golden("vidtome")
    N, C = 192, 32
    rng = np.random.default_rng(42)
    bank = None
    for ci, F in enumerate([4, 4, 3, 1, 2]):
        x = torch.from_numpy(rng.standard_normal((2 * F, N, C)).astype(np.float32))
        r = OV.compute_merge(x, F, bank, int(g[f"c{ci}_randf"]), float(g[f"c{ci}_coin"]))
        assert r["merged"].shape[1] == int(g[f"c{ci}_T"])
        assert np.array_equal(r["merged"][:, ::7, ::5].numpy(), g[f"c{ci}_merged"])       # pure gathers: bit-exact
        assert np.array_equal(r["unm"].numpy(), g[f"c{ci}_unm"])
        bank = r["bank_new"]
        assert bank.shape[1] == int(g[f"c{ci}_bank_T"])
        assert np.array_equal(bank[:, ::7, ::5].numpy(), g[f"c{ci}_bank"])
        # gather codes reproduce merged from (x, old bank)
        ids = torch.arange(r["merged"].shape[1], dtype=torch.float32)[None, :, None].repeat(2, 1, 1)
        assert torch.equal(r["unmerge"](ids)[:F].reshape(-1).long(), r["unm"])


def _chain(g, tag, frames, seed, align_batch):
    N, C = 192, 32
    rng = np.random.default_rng(1000 + seed)
    bank = None
    for ci, F in enumerate(frames):
        x = torch.from_numpy(rng.standard_normal((2 * F, N, C)).astype(np.float32))
        r = OV.compute_merge(x, F, bank, [int(v) for v in g[f"{tag}{ci}_randf"]], float(g[f"{tag}{ci}_coin"]), align_batch=align_batch)
        assert r["merged"].shape[1] == int(g[f"{tag}{ci}_T"])
        assert np.array_equal(r["merged"][:, ::7, ::5].numpy(), g[f"{tag}{ci}_merged"])     # pure gathers: bit-exact
        unm = r["unm"] if r["unm"].dim() == 2 else r["unm"][None].expand(2, -1)
        assert np.array_equal(unm.numpy(), g[f"{tag}{ci}_unm"])
        bank = r["bank_new"]
        assert bank.shape[1] == int(g[f"{tag}{ci}_bank_T"])
        assert np.array_equal(bank[:, ::7, ::5].numpy(), g[f"{tag}{ci}_bank"])
        yield F, r


def test_vidtome_multi_round_local_merge(golden):
    """patch.py:43-56: chunks longer than target_stride merge in several randframe rounds (8 -> 2 -> 1, 16 -> 4 -> 1), the unmerged tokens of
    the earlier rounds joining the dst set -- the reference's own compute_merge run on 8 / 8 / 6 / 4 / 16-frame chunks."""
    rounds = [r["rounds"] for _, r in _chain(golden("vidtome"), "m", [8, 8, 6, 4, 16], 5, True)]
    assert rounds == [2, 2, 1, 1, 2]


def test_vidtome_per_sample_matching(golden):
    """merge.py:109-118 / align_batch=False: every batch entry gets its own matching."""
    differ = 0
    for F, r in _chain(golden("vidtome"), "p", [4, 4, 3, 1, 8], 6, False):
        if F > 1:
            assert r["unm"].dim() == 2
            differ += int(not torch.equal(r["unm"][0], r["unm"][1]))
    assert differ >= 3


def test_chunks_windows_fusion(golden):
    g = golden("pipeline")
    for tag, flen in {"n8": 8, "n30": 30, "n300": 300, "w120": 120, "n3": 3}.items():
        rf, fl = g[f"chunks_{tag}_draws"]
        ch = OP.get_chunks(flen, 4, int(rf), float(fl), torch.from_numpy(g[f"chunks_{tag}_perm"]))
        assert np.array_equal(torch.cat(ch).numpy(), g[f"chunks_{tag}_flat"])
        assert [len(c) for c in ch] == list(g[f"chunks_{tag}_lens"])
        assert OP.n_chunks(flen, 4, int(rf)) == len(ch)
    assert OP.temporal_windows(300, 64) == ([0, 59, 118, 177, 236], [5, 5, 5, 5])
    for N in (8, 30, 64, 65, 127, 300):
        h, w = 3, 6
        gg = np.random.default_rng(N)
        x = torch.from_numpy(gg.standard_normal((N, 4, h, w)).astype(np.float32))
        noises = torch.from_numpy(gg.standard_normal((N, 4, h, w)).astype(np.float32))
        nt, nf = OP.temporal_denoise(x, x * 2 + 1, 0.01 * 0.3, noises, 64, [torch.arange(0, 2), torch.arange(2, w)],
                                     lambda xt, ct, ch, sl: xt * 0.5 + ct * 0.25 + (sl + 1) * 0.01)
        np.testing.assert_allclose(nt.numpy(), g[f"tden_{N}_nt"], atol=1e-6)
        np.testing.assert_allclose(nf.numpy(), g[f"tden_{N}_nf"], atol=1e-6)
        starts, _ = OP.temporal_windows(N, 64)
        assert starts == sorted({int(s) for s, _ in g[f"tden_{N}_windows"]})
    np.testing.assert_allclose(OP.alpha_schedule(0.01, 0.01, 20), g["ddim_alphas"], rtol=1e-12)
    gg = np.random.default_rng(9)
    x = torch.from_numpy(gg.standard_normal((3, 4, 5, 6)).astype(np.float32))
    inp = torch.cat([x, x])
    eps = torch.cat([inp[:3] * 0.3, inp[3:] * 0.7 + 1.0])
    np.testing.assert_allclose(OP.cfg(eps, 2.0).numpy(), g["pn_out"], atol=1e-6)
