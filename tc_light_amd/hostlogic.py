"""Integer host logic of the denoising loop: chunking, yt-plane windows, decay schedule (no tensors on the data path).

get_chunks        utils/VidToMe/generate_utils.py:174-205
temporal_windows  generate.py:246-260
alpha_schedule    generate.py:228-229
The reference draws from the global numpy / torch CPU RNGs; here the draws come from explicit seeded streams so that
"identical seeds" means the same choices on any device (hard part 3).
"""
import math

import numpy as np
import torch


def chunks_from_draws(flen, chunk_size, rand_first, flip_draw, perm, merge_global=True, chunk_ord="mix", perm_div=4.0):
    """Pure function of the three draws: rand_first in [0, chunk_size), flip_draw in [0,1), perm = permutation of chunk ids."""
    first = rand_first + 1
    bounds = [(0, min(first, flen))]
    s = first
    while s < flen:
        bounds.append((s, min(s + chunk_size, flen)))
        s += chunk_size
    chunks = [list(range(a, b)) for a, b in bounds]
    if flip_draw > 0.5:
        chunks = chunks[::-1]
    if not merge_global:
        return chunks
    order = [int(p) for p in perm] if chunk_ord in ("rand", "mix") else list(range(len(chunks)))
    if chunk_ord == "mix":
        # the first len/perm_div draws stay in drawn order; the remaining chunks follow in index order, walked from whichever end lies
        # nearer (in chunk index) to the last drawn one (generate_utils.py:189-201)
        head = order[:int(len(order) / perm_div)]
        tail = sorted(order[len(head):])
        if head and tail and abs(tail[-1] - head[-1]) < abs(tail[0] - head[-1]):
            tail.reverse()
        order = head + tail
    return [chunks[i] for i in order]


def n_chunks(flen, chunk_size, rand_first):
    first = rand_first + 1
    return 1 + max(0, math.ceil((flen - first) / chunk_size))


class ChunkSampler:
    """Seeded stand-in for the np.random / torch.randperm calls of get_chunks."""

    def __init__(self, seed, chunk_size=4, merge_global=True, chunk_ord="mix-4"):
        self.np_rng = np.random.RandomState(seed)
        self.t_gen = torch.Generator(device="cpu").manual_seed(int(seed))
        self.chunk_size, self.merge_global = chunk_size, merge_global
        self.perm_div = float(chunk_ord.split("-")[-1]) if "-" in chunk_ord else 3.0
        self.chunk_ord = "mix" if "mix" in chunk_ord else chunk_ord

    def get_chunks(self, flen):
        rf = int(self.np_rng.randint(0, self.chunk_size))
        fl = float(self.np_rng.rand())
        n = n_chunks(flen, self.chunk_size, rf)
        perm = torch.randperm(n, generator=self.t_gen) if (self.merge_global and self.chunk_ord in ("rand", "mix")) else torch.arange(n)
        return chunks_from_draws(flen, self.chunk_size, rf, fl, perm, self.merge_global, self.chunk_ord, self.perm_div)


def temporal_windows(n, win):
    """-> (window starts, overlaps): n_slices = ceil((n-1)/(win-1)); later windows overwrite then scale the overlap."""
    n_slices = math.ceil((n - 1) / (win - 1)) if n > 1 else 1
    if n_slices > 1:
        total = n_slices * win - n
        ov = total // (n_slices - 1)
        last = ov + total % (n_slices - 1)
        ovl = [ov] * (n_slices - 2) + [last]
        cs = np.cumsum(ovl)
        return [0] + [int((i + 1) * win - cs[i]) for i in range(n_slices - 1)], [int(o) for o in ovl]
    return [0], [0]


def alpha_schedule(alpha_t, final_factor_t, n_steps):
    return [alpha_t * final_factor_t ** min(i / n_steps, 1) for i in range(n_steps)]


def shard_range(n, rank, world):
    """Contiguous frame block of `rank` (sizes differ by at most one; SURVEY 8(d) config 3: 38/38/38/38/37/37/37/37)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def expon_lr(step, lr_init, lr_final, max_steps):
    """get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, max_steps=max_steps)(step)  (utils/general_utils.py:31-64, delay off)."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
