"""GPU: the multi-GPU form of stage 1 / stage 2 on the HIP kernels.

1. tcl_exposure_grad / tcl_unique_tensor_grad: the partial gradients and partial losses of a mini-batch dealt to 2 and 3 "ranks" (computed
   one after the other on this GPU) add up to the whole-batch gradient / loss of the single-GPU path.
2. Two real processes (gloo, both on cuda:0 -- the box has one GPU; RCCL needs one GPU per rank) run post_opt.exposure_align /
   unique_tensor_optimization with dist=Dist(rank, 2): same losses / parameters as the single-process whole-stage drivers, which
   test_gpu_path2.py pins against the reference goldens.
Tolerances: float-atomic sums -> 2e-5 relative on losses, 5e-5 absolute on parameters except Adam(eps 1e-15) noise rows.
"""
import os
import socket

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tc_light_amd import post_opt
    return post_opt


def test_partial_gradients_add_up():
    P = _setup()
    from tc_light_amd.lib import lib, stream
    from tc_light_amd.parallel import deal_slots
    L = lib()
    n, h, w = 5, 176, 192
    d = synth.video_clip(n, h, w, seed=11)
    inv, k = synth.track_ids(n, h, w, seed=3)
    dev = "cuda"
    ed, fl, mk = (d[x].to(dev).contiguous() for x in ("edited", "past_flows", "masks"))
    inv = inv.to(device=dev, dtype=torch.int32)
    row = [3, 0, 4, 1, 2]
    fsh = P.OptDataset(d["edited"], d["past_flows"], d["masks"], device=dev).flow_shift
    feat = torch.empty(3, k, device=dev)
    L.tcl_scatter_mean_rgb2sh(ed, inv, feat, torch.empty(k, device=dev), n, h, w, k, 1, stream())
    expo = (torch.eye(3, 4, device=dev)[None].repeat(n, 1, 1) + 0.02 * torch.randn(n, 3, 4, device=dev)).contiguous()

    def run(world, stage):
        g = torch.zeros(3 * k if stage == 2 else n * 12, device=dev)
        loss = torch.zeros(world, device=dev)
        for r in range(world):
            slots, b_glob, nvalid = deal_slots(row, r, world)
            cat = torch.tensor(slots + [max(s - 1, 0) for s in slots], dtype=torch.int32, device=dev)
            ws = torch.empty(L.tcl_stage_workspace_bytes(len(slots), h, w), dtype=torch.uint8, device=dev)
            if stage == 2:
                L.tcl_unique_tensor_grad(ed, fl, mk, fsh, inv, n, h, w, k, 1, cat, len(slots), b_glob, nvalid, 0.2, 0.8, 0.05, feat, g, loss[r:r + 1], ws, stream())
            else:
                L.tcl_exposure_grad(ed, fl, mk, fsh, n, h, w, cat, len(slots), b_glob, nvalid, 0.2, 0.8, expo, g, loss[r:r + 1], ws, stream())
        return g.cpu(), float(loss.sum())

    for stage in (1, 2):
        g1, l1 = run(1, stage)
        for world in (2, 3):
            gw, lw = run(world, stage)
            assert abs(lw - l1) < 2e-5 * abs(l1), (stage, world, lw, l1)
            assert (gw - g1).abs().max() < 2e-5 * g1.abs().max(), (stage, world)
        assert g1.abs().max() > 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    n, h, w = 6, 176, 192
    d = synth.video_clip(n, h, w, seed=11)
    inv, k = synth.track_ids(n, h, w, seed=3)
    s1 = np.array([[2, 1, 5, 3], [4, 2, -1, -1], [1, 3, 5, 4], [2, 5, 1, -1]], np.int32)       # frame 0 never "current" (conditioning note)
    s2 = np.array([[0, 4, 2, 5], [3, 1, -1, -1], [5, 0, 1, 3], [4, 2, -1, -1]], np.int32)
    return d, inv, k, s1, s2


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tc_light_amd import post_opt as P
    from tc_light_amd.parallel import Dist
    d, inv, k, s1, s2 = _inputs()
    pd = Dist(rank, world)
    ds = P.OptDataset(d["edited"], d["past_flows"], d["masks"], device="cuda")
    _, expo, l1 = P.exposure_align(ds, s1, epochs=2, batch_size=4, iters_per_epoch=2, dist=pd)
    ds2 = P.OptDataset(d["edited"], d["past_flows"], d["masks"], device="cuda")     # fresh targets: stage 1's frame-0 exposure is
    out, feat, l2 = P.unique_tensor_optimization(ds2, inv.cuda(), s2, batch_size=4, k=k, dist=pd)   # rounding-noise driven
    torch.cuda.synchronize()
    if rank == 0:
        ret.put(tuple(t.detach().cpu().numpy() for t in (expo, l1, out, feat, l2)))      # numpy: no fd-sharing race with the exiting worker
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_global_stages_equal_single_process():
    P = _setup()
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    expo, l1, out, feat, l2 = (torch.from_numpy(a) for a in ret.get())
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    d, inv, k, s1, s2 = _inputs()
    ds = P.OptDataset(d["edited"], d["past_flows"], d["masks"], device="cuda")
    _, expo_r, l1_r = P.exposure_align(ds, s1, epochs=2, batch_size=4, iters_per_epoch=2)
    ds2 = P.OptDataset(d["edited"], d["past_flows"], d["masks"], device="cuda")
    out_r, feat_r, l2_r = P.unique_tensor_optimization(ds2, inv.cuda(), s2, batch_size=4, k=k)
    np.testing.assert_allclose(l1.numpy(), l1_r.cpu().numpy(), rtol=2e-5)
    assert (expo[1:] - expo_r.cpu()[1:]).abs().max() < 5e-5
    np.testing.assert_allclose(l2.numpy(), l2_r.cpu().numpy(), rtol=2e-5)
    diff = (out - out_r.cpu()).abs()
    assert (diff > 5e-5).float().mean() < 2e-3, (diff > 5e-5).float().mean()
    assert (feat - feat_r.cpu()).abs().median() < 1e-6
