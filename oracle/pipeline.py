"""Oracle (CPU) for the path-1 host loop: chunking, yt-plane windows, noise fusion, CFG.
TEST INFRASTRUCTURE -- see oracle/__init__.py.  Pinned by tests/golden/pipeline.npz (reference methods run in place)."""
import math

import numpy as np
import torch


def get_chunks(flen, chunk_size, rand_first, flip_draw, perm, merge_global=True, chunk_ord="mix", perm_div=4.0):
    """utils/VidToMe/generate_utils.py:174-205 with the three RNG draws made explicit:
    rand_first = np.random.randint(0, chunk_size), flip_draw = np.random.rand(), perm = torch.randperm(n_chunks)."""
    idx = torch.arange(flen)
    first = rand_first + 1
    rest = idx[first:].split(chunk_size)
    chunks = [idx[:first]] + list(rest) if len(rest[0]) > 0 else [idx[:first]]
    if flip_draw > 0.5:
        chunks = chunks[::-1]
    if not merge_global:
        return chunks
    if chunk_ord == "rand":
        order = list(perm)
    elif chunk_ord == "mix":
        randord = [int(p) for p in perm]
        rand_len = int(len(randord) / perm_div)
        seqord = sorted(randord[rand_len:])
        if rand_len > 0:
            randord = randord[:rand_len]
            if abs(seqord[-1] - randord[-1]) < abs(seqord[0] - randord[-1]):
                seqord = seqord[::-1]
            order = randord + seqord
        else:
            order = seqord
    else:
        order = list(range(len(chunks)))
    return [chunks[i] for i in order]


def n_chunks(flen, chunk_size, rand_first):
    first = rand_first + 1
    return 1 + (max(flen - first, 0) + chunk_size - 1) // chunk_size


def temporal_windows(n, win):
    """generate.py:246-260 -> (window starts, overlap list)."""
    n_slices = math.ceil((n - 1) / (win - 1))
    if n_slices > 1:
        total = n_slices * win - n
        ov = total // (n_slices - 1)
        last = ov + total % (n_slices - 1)
        ovl = [ov] * (n_slices - 2) + [last]
        cs = np.cumsum(ovl)
        return [0] + [int((i + 1) * win - cs[i]) for i in range(n_slices - 1)], [int(o) for o in ovl]
    return [0], [0]


def adain(content, style, eps=1e-5):
    """utils/general_utils.py:137-156."""
    def ms(f):
        n, c = f.shape[:2]
        return f.reshape(n, c, -1).mean(2).view(n, c, 1, 1), (f.reshape(n, c, -1).var(2) + eps).sqrt().view(n, c, 1, 1)
    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm


def temporal_denoise(x, concat_conds, alpha_t, noises, win, chunks, pred_noise):
    """generate.py:241-284.  pred_noise(xt [w',c,n,h], cc_t, chunk, sl_i) -> same shape."""
    starts, ovl = temporal_windows(len(x), win)
    nt = torch.zeros_like(x)
    for k, sl in enumerate(starts):
        for ch in chunks:
            xt = x[sl:sl + win][:, :, :, ch].permute(3, 1, 0, 2)
            ct = concat_conds[sl:sl + win][:, :, :, ch].permute(3, 1, 0, 2)
            pred = pred_noise(xt, ct, ch, sl)
            nt[sl:sl + win, :, :, ch] = pred.permute(2, 1, 3, 0)
        if sl > 0:
            nt[sl:sl + ovl[k - 1]] *= np.sqrt(0.5)
    nt = adain(nt, noises)
    return nt, (alpha_t ** 0.5) * nt + ((1 - alpha_t) ** 0.5) * noises


def alpha_schedule(alpha_t, final_factor_t, n_steps):
    """generate.py:228-229."""
    return [alpha_t * final_factor_t ** min(i / n_steps, 1) for i in range(n_steps)]


def cfg(eps, guidance):
    """generate.py:349-350."""
    u, c = eps.chunk(2)
    return u + guidance * (c - u)
