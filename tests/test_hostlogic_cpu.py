"""CPU: the product's host logic (tc_light_amd/hostlogic.py) against the goldens produced by the reference's own code."""
import numpy as np
import torch

from tc_light_amd import hostlogic as HL


def test_chunks_match_reference(golden):
    g = golden("pipeline")
    for tag, flen in {"n8": 8, "n30": 30, "n300": 300, "w120": 120, "n3": 3}.items():
        rf, fl = g[f"chunks_{tag}_draws"]
        ch = HL.chunks_from_draws(flen, 4, int(rf), float(fl), torch.from_numpy(g[f"chunks_{tag}_perm"]))
        assert [c for chunk in ch for c in chunk] == list(g[f"chunks_{tag}_flat"])
        assert [len(c) for c in ch] == list(g[f"chunks_{tag}_lens"])
        assert HL.n_chunks(flen, 4, int(rf)) == len(ch)


def test_chunk_sampler_covers_everything():
    s = HL.ChunkSampler(12345)
    for flen in (1, 2, 5, 30, 38, 120, 160):
        for _ in range(5):
            ch = s.get_chunks(flen)
            flat = sorted(c for chunk in ch for c in chunk)
            assert flat == list(range(flen)) and all(1 <= len(c) <= 4 for c in ch)


def test_windows_alpha_shards(golden):
    g = golden("pipeline")
    assert HL.temporal_windows(300, 64) == ([0, 59, 118, 177, 236], [5, 5, 5, 5])
    assert HL.temporal_windows(30, 64) == ([0], [0]) and HL.temporal_windows(64, 64) == ([0], [0])
    for n in (8, 30, 64, 65, 127, 300):
        starts, _ = HL.temporal_windows(n, 64)
        assert starts == sorted({int(s) for s, _ in g[f"tden_{n}_windows"]})
    np.testing.assert_allclose(HL.alpha_schedule(0.01, 0.01, 20), g["ddim_alphas"], rtol=1e-12)
    sizes = [HL.shard_range(300, r, 8) for r in range(8)]
    assert [b - a for a, b in sizes] == [38, 38, 38, 38, 37, 37, 37, 37] and sizes[0][0] == 0 and sizes[-1][1] == 300


def test_shard_track_ids_equal_fresh_flowid():
    """Multi-GPU stage 2 runs on each rank's frame block with the GLOBAL track ids restricted to the block and renumbered densely
    (generate.py of this repo, `shard_post_opt`).  That is the partition get_flowid (flow_utils.py:56-93) gives when started at the
    block's first frame -- i.e. the reference run on the shard."""
    import torch
    import synth
    from oracle import path2 as O2
    n, h, w = 7, 40, 56
    d = synth.video_clip(n, h, w, seed=4)
    frames = d["frames"].clone()
    frames[:, :, 0, 0] = 1.0                                   # same max in every block -> same colour threshold
    fwd = torch.zeros(n, 2, h, w)
    fwd[:, 0], fwd[:, 1] = -1.5, -0.5                          # forward flow of the synthetic translation
    fwd += 0.05 * torch.randn(fwd.shape, generator=torch.Generator().manual_seed(1))
    masks = d["masks"]
    ids = O2.get_flowid(frames, fwd, masks)
    for lo, hi in ((0, 4), (4, 7), (2, 6)):
        _, local = torch.unique(ids[lo:hi].reshape(-1), return_inverse=True)
        fresh = O2.get_flowid(frames[lo:hi], fwd[lo:hi], masks[lo:hi]).reshape(-1)
        # same partition <=> the pairing (local id, fresh id) is one-to-one
        pairs = torch.unique(torch.stack([local, fresh], 1), dim=0)
        assert pairs.shape[0] == local.max().item() + 1 == fresh.max().item() + 1


def test_bench_gpus_n_without_gpu_fails_loudly():
    """`python bench.py --gpus 2` on a box without GPUs must say so (exit != 0, one clear line) -- never run a silent 1-rank or CPU pass."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs a GPU" in (p.stderr + p.stdout)


def test_skewed_pair_schedule_and_group_cut(monkeypatch):
    """TCL_SKEW, host side only: `Generator._run_groups` cuts a lone group at the chunk boundary nearest to half its frames and pairs
    the groups; `UNetEngine.forward_pair` takes the draws for A then B and alternates the two UNet generators with A two segments ahead, so per
    merging block the issue order is  M(A,b) .. A(A,b) | M(B,b) | F(A,b) P(A,b+1) | A(B,b) | M(A,b+1)  -- bank order A before B in every block."""
    from types import SimpleNamespace
    from tc_light_amd.generate import Generator
    from tc_light_amd.unet import UNetEngine
    log = []

    class Eng:
        def __init__(self):
            self.tome = SimpleNamespace(begin_step=lambda Fs, size: log.append(("draws", tuple(Fs))) or [(F, [0], 0.5) for F in Fs])

        def _forward_gen(self, x, Fs, Hh, Ww, t, text, cfg_pair=False, skew=False, chunks=None):
            assert skew and chunks is not None and len(chunks) == len(Fs)
            for b in range(3):                             # three merging blocks
                log.append((x, "P", b)); yield "M"
                log.append((x, "M", b)); yield "A"
                log.append((x, "A", b)); yield "F"
                log.append((x, "F", b))
            return x + "-eps"

    ea, eb = UNetEngine.forward_pair(Eng(), "a", [4, 4], "b", [4, 3], 8, 8, 1.0, None, cfg_pair=True)
    assert (ea, eb) == ("a-eps", "b-eps")
    assert log[:2] == [("draws", (4, 4)), ("draws", (4, 3))]
    seq = [e for e in log[2:]]
    for b in range(3):                                     # chains: A's before B's in every block; a chain is issued right behind the OTHER group's attn1
        assert seq.index(("a", "M", b)) < seq.index(("b", "M", b))
        assert seq[seq.index(("b", "M", b)) - 1] == ("a", "A", b)
        if b:
            assert seq[seq.index(("a", "M", b)) - 1] == ("b", "A", b - 1)
        # between a chain and its own attn1 the main stream gets the other group's non-attention work
        between = seq[seq.index(("b", "M", b)) + 1:seq.index(("b", "A", b))]
        assert ("a", "F", b) in between
    assert sorted(seq) == sorted((g, s, b) for g in "ab" for s in "PMAF" for b in range(3))

    calls = []
    unet = SimpleNamespace(forward_pair=lambda xa, Fa, xb, Fb, *r, **k: calls.append(("pair", Fa, Fb)) or ("ea", "eb"),
                           forward_many=lambda x, Fs, *r, **k: calls.append(("one", Fs)) or "e")
    gen = SimpleNamespace(unet=unet)
    chunks = [list(range(3)), list(range(3, 19)), list(range(19, 35)), list(range(35, 40))]      # 3 + 16 + 16 + 5 frames
    pack = lambda grp: (None, None, sum(len(c) for c in grp))
    monkeypatch.setenv("TCL_SKEW", "1")
    Generator._run_groups(gen, [chunks], 8, 8, 1.0, None, pack, lambda *a: None)
    assert calls == [("pair", [3, 16], [16, 5])]
    calls.clear()
    Generator._run_groups(gen, [chunks[:2], chunks[2:3], chunks[3:]], 8, 8, 1.0, None, pack, lambda *a: None)     # cut -> [3] [16] [16] [5]: two pairs
    assert calls == [("pair", [3], [16]), ("pair", [16], [5])]
    calls.clear()
    monkeypatch.setenv("TCL_SKEW", "cut")
    Generator._run_groups(gen, [chunks], 8, 8, 1.0, None, pack, lambda *a: None)
    assert calls == [("one", [3, 16]), ("one", [16, 5])]
    calls.clear()
    monkeypatch.setenv("TCL_SKEW", "0")
    Generator._run_groups(gen, [chunks], 8, 8, 1.0, None, pack, lambda *a: None)
    assert calls == [("one", [3, 16, 16, 5])]
