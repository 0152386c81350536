"""CPU, gloo, world_size 2: the multi-GPU choreography of tc_light_amd/parallel.py assembles exactly what one process computes."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _items(n_total, w, win=64):
    from tc_light_amd import hostlogic as HL
    starts, ovl = HL.temporal_windows(n_total, win)
    cols = [list(range(i, min(i + 4, w))) for i in range(0, w, 4)]
    items = []
    for k, sl in enumerate(starts):
        nwin = min(win, n_total - sl)
        nkeep = (starts[k + 1] - sl) if k + 1 < len(starts) else nwin
        for ch in cols:
            items.append((sl, nwin, ch, sl + ovl[k - 1] if k > 0 else 0, nkeep))
    return items


def _compute(x_full, cc_full, items, out):
    """Stand-in for the yt-plane UNet calls of one window: a deterministic function of the window's frames and each chunk's columns."""
    assert len({it[:2] for it in items}) == 1           # one window per call
    for item in items:
        _compute_one(x_full, cc_full, item, out)


def _compute_one(x_full, cc_full, item, out):
    sl, nwin, cols, up, nkeep = item
    blk = x_full[sl:sl + nwin][:, :, :, cols] * 0.5 + cc_full[sl:sl + nwin][:, :, :, cols] * 0.25 + (sl + 1) * 0.01
    scale = torch.ones(nwin, 1, 1, 1)
    for i in range(nwin):
        if sl + i < up:
            scale[i] = 0.5 ** 0.5
    out[sl:sl + nkeep][:, :, :, cols] = (blk * scale)[:nkeep]


def _worker(rank, world, port, n_total, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tc_light_amd.parallel import Dist, sharded_temporal_pass
    d = Dist(rank, world)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n_total, 4, 3, 10, generator=g)
    cc = torch.randn(n_total, 4, 3, 10, generator=g)
    lo, hi = d.range(n_total)
    cc_full = d.gather_frames(cc[lo:hi].clone(), n_total)
    assert torch.equal(cc_full, cc)
    nt = sharded_temporal_pass(d, x[lo:hi].clone(), cc_full, n_total, _items(n_total, 10), _compute)
    full = d.gather_frames(nt, n_total)
    if rank == 0:
        ret.put(full)
    assert d.max_float(float(rank), "cpu") == world - 1
    dist.destroy_process_group()


def test_sharded_temporal_pass_world2():
    for n_total in (9, 70):          # one window / two overlapping windows, uneven shards
        ctx = mp.get_context("spawn")
        ret = ctx.SimpleQueue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, ret)) for r in range(2)]
        for p in procs:
            p.start()
        got = ret.get()
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        # single-process result in the reference's sequential order (later windows overwrite the overlap)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(n_total, 4, 3, 10, generator=g)
        cc = torch.randn(n_total, 4, 3, 10, generator=g)
        ref = torch.zeros_like(x)
        for it in _items(n_total, 10):
            sl, nwin, cols, up, nkeep = it
            _compute_one(x, cc, (sl, nwin, cols, up, nwin), ref)     # write ALL frames, sequentially, like generate.py:265-278
        assert torch.equal(got, ref)
