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
