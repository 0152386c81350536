"""Deterministic synthetic inputs shared by the golden generator, the parity tests and bench.py.

Only numpy's Generator (PCG64, version-stable) is used so that the same seed gives the same
arrays in the build container (where the goldens are made) and on the GPU box.
"""
import numpy as np
import torch


def _smooth(rng, shape, passes=3):
    x = rng.random(shape, dtype=np.float32)
    for _ in range(passes):  # cheap separable box blur -> low-pass content
        x = (np.roll(x, 1, -1) + x + np.roll(x, -1, -1)) / 3
        x = (np.roll(x, 1, -2) + x + np.roll(x, -1, -2)) / 3
    return x


def video_clip(n, h, w, seed=12345, shift=(1.5, 0.5), jitter=0.03):
    """Frames = one low-pass base image translated by k*shift px (+ per-frame gain/bias jitter
    and a little noise), the matching analytic backward flow (+noise) and a soft mask.
    Returns dict of float32 torch tensors: edited [n,3,h,w], past_flows [n,2,h,w], masks [n,1,h,w]."""
    rng = np.random.default_rng(seed)
    pad = int(max(abs(shift[0]), abs(shift[1])) * n) + 4
    base = _smooth(rng, (3, h + 2 * pad, w + 2 * pad), passes=4)
    base = (base - base.min()) / (base.max() - base.min())
    frames = np.empty((n, 3, h, w), np.float32)
    for k in range(n):
        dx, dy = shift[0] * k, shift[1] * k
        x0, y0 = int(np.floor(dx)), int(np.floor(dy))
        fx, fy = dx - x0, dy - y0
        def crop(ox, oy):
            return base[:, pad + y0 + oy: pad + y0 + oy + h, pad + x0 + ox: pad + x0 + ox + w]
        frames[k] = ((1 - fx) * (1 - fy) * crop(0, 0) + fx * (1 - fy) * crop(1, 0)
                     + (1 - fx) * fy * crop(0, 1) + fx * fy * crop(1, 1))
    gain = 1 + jitter * rng.standard_normal((n, 3, 1, 1)).astype(np.float32)
    bias = jitter * rng.standard_normal((n, 3, 1, 1)).astype(np.float32)
    edited = np.clip(frames * gain + bias + 0.01 * rng.standard_normal(frames.shape).astype(np.float32), 0, 1)
    flow = np.empty((n, 2, h, w), np.float32)
    # frame k(x) = base(x + k*shift)  =>  frame k (x) = frame k-1 (x + shift): backward flow = +shift
    flow[:, 0] = shift[0]
    flow[:, 1] = shift[1]
    flow += 0.05 * rng.standard_normal(flow.shape).astype(np.float32)
    flow[0] = 0
    m = _smooth(rng, (n, 1, h, w), passes=6)
    m = (m > np.quantile(m, 0.1)).astype(np.float32)
    m = np.clip(m * (0.9 + 0.1 * rng.random(m.shape, dtype=np.float32)), 0, 1)
    return dict(frames=torch.from_numpy(frames), edited=torch.from_numpy(edited.astype(np.float32)),
                past_flows=torch.from_numpy(flow), masks=torch.from_numpy(m))


def batches(n, batch_size, epochs, seed=7):
    """Explicit mini-batch index sequence standing in for DataLoader(shuffle=True)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(epochs):
        p = rng.permutation(n)
        for i in range(0, n, batch_size):
            out.append(torch.from_numpy(p[i:i + batch_size].astype(np.int64)))
    return out


def track_ids(n, h, w, seed=3, reuse=0.7):
    """A plausible unq_inv [n*h*w] int64: ids of frame k mostly re-use frame k-1 ids shifted by one pixel."""
    rng = np.random.default_rng(seed)
    ids = np.empty((n, h, w), np.int64)
    ids[0] = np.arange(h * w).reshape(h, w)
    last = h * w
    for k in range(1, n):
        prev = np.roll(ids[k - 1], 1, axis=1)
        fresh = rng.random((h, w)) > reuse
        fresh[:, 0] = True
        cnt = int(fresh.sum())
        cur = prev.copy()
        cur[fresh] = last + np.arange(cnt)
        last += cnt
        ids[k] = cur
    return torch.from_numpy(ids.reshape(-1)), last
