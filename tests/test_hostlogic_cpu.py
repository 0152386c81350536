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
