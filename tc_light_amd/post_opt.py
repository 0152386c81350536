"""Path 2 host side: the reference's stage-1/2 operator seam on top of the HIP C ABI.

Same names / argument meaning as the reference (SURVEY 8(b) "Stage-1/2 ops"):
  warp_flow(frames, past_flows)                    utils/flow_utils.py:5
  relaxed_ms_ssim(X, Y, data_range=1, start_level=1)   utils/loss_utils.py:125
  TVLoss(weight)(x)                                utils/loss_utils.py:324
  OptDataset                                       utils/dataloader.py:9
  exposure_align / unique_tensor_optimization      generate.py:354 / :453
The three loss ops are torch.autograd.Functions whose forward AND backward are HIP kernels.
"""
import os

import numpy as np
import torch

from .lib import lib, stream, check

F32 = torch.float32


class _WarpFlow(torch.autograd.Function):
    @staticmethod
    def forward(ctx, frames, past_flows):
        frames = check(frames.contiguous(), F32)
        flows = check(past_flows.contiguous(), F32)
        n, c, h, w = frames.shape
        out = torch.empty_like(frames)
        lib().tcl_warp_flow_fwd(frames, flows, out, n, c, h, w, flows.shape[1], stream())
        ctx.save_for_backward(flows)
        return out

    @staticmethod
    def backward(ctx, gout):
        (flows,) = ctx.saved_tensors
        gout = check(gout.contiguous(), F32)
        n, c, h, w = gout.shape
        gimg = torch.empty_like(gout)
        lib().tcl_warp_flow_bwd(gout, flows, gimg, n, c, h, w, flows.shape[1], stream())
        return gimg, None


def warp_flow(frames, past_flows):
    return _WarpFlow.apply(frames, past_flows)


class _MsSsim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = check(x.contiguous(), F32), check(y.contiguous(), F32)
        b, c, h, w = x.shape
        ws = torch.empty(lib().tcl_msssim_workspace_bytes(b * c, h, w), dtype=torch.uint8, device=x.device)
        val = torch.empty(1, dtype=F32, device=x.device)
        grad = torch.empty_like(x) if x.requires_grad else None
        lib().tcl_ms_ssim_loss(x, y, b * c, h, w, val, grad if grad is not None else 0, ws, stream())
        ctx.grad = grad
        return 1.0 - val[0]

    @staticmethod
    def backward(ctx, g):
        return (-g * ctx.grad if ctx.grad is not None else None), None


def relaxed_ms_ssim(X, Y, data_range=1, start_level=1):
    if data_range != 1 or start_level != 1:
        raise NotImplementedError("HIP relaxed_ms_ssim implements the configuration TC-Light uses "
                                  "(data_range=1, start_level=1; generate.py:416,510)")
    if X.shape != Y.shape or X.dim() != 4:
        raise ValueError(f"Input images should have the same 4-d dimensions, but got {X.shape} and {Y.shape}.")
    assert min(X.shape[-2:]) > 160, "Image size should be larger than 160 due to the 4 downsamplings in ms-ssim"
    return _MsSsim.apply(X, Y)


class _TV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        x = check(x.contiguous(), F32)
        b, c, h, w = x.shape
        val = torch.empty(1, dtype=F32, device=x.device)
        grad = torch.empty_like(x)
        ws = torch.empty(512, dtype=F32, device=x.device)          # 2 KiB: fixed-point loss accumulators
        lib().tcl_tv_loss(x, b, c, h, w, float(weight), val, grad, ws, stream())
        ctx.grad = grad
        return val[0]

    @staticmethod
    def backward(ctx, g):
        return g * ctx.grad, None


class TVLoss(torch.nn.Module):
    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        return _TV.apply(x, self.TVLoss_weight)


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


class OptDataset:
    """utils/dataloader.py:9-42 -- device-resident stage-1/2 inputs."""

    def __init__(self, edited_images, past_flows, mask_bwd, device, dtype=F32):
        self.edited_images = edited_images.to(dtype=dtype, device=device).contiguous()
        self.past_flows = past_flows.to(dtype=dtype, device=device).contiguous()
        self.mask_bwd = mask_bwd.to(dtype=dtype, device=device).contiguous()
        if self.edited_images.max() > 1:
            self.edited_images = self.edited_images / 255.0
        self.device, self.dtype = device, dtype
        self._flow_shift = None

    @property
    def flow_shift(self):
        """int32 [N] on the device: per-frame fixed-point exponent of the flow term's gradient scatter (csrc/path2.hip FX_FLOW_SHIFT), from the
        frames' flows and masks -- computed on first use, once per clip."""
        if self._flow_shift is None:
            n, _, h, w = self.past_flows.shape
            dev = self.past_flows.device
            self._flow_shift = torch.empty(n, dtype=torch.int32, device=dev)
            scratch = torch.empty(h * w + 1, dtype=torch.int32, device=dev)
            lib().tcl_flow_cell_shift(self.past_flows, self.mask_bwd, n, h, w, scratch, self._flow_shift, stream())
        return self._flow_shift

    def __len__(self):
        return len(self.edited_images)


def make_schedule(n, batch_size, epochs, rng):
    """Stand-in for DataLoader(shuffle=True): host int32 [iters, batch] (-1 padded) from a numpy Generator."""
    rows = []
    for _ in range(epochs):
        p = rng.permutation(n)
        for i in range(0, n, batch_size):
            r = np.full(batch_size, -1, np.int32)
            c = p[i:i + batch_size]
            r[:len(c)] = c
            rows.append(r)
    return np.stack(rows)


def _pack_schedule(batches, batch_size, device):
    """list of index tensors / [iters,batch] array -> (host sched int32, device cat int32 [iters, 2*batch])."""
    if isinstance(batches, np.ndarray):
        sched = np.ascontiguousarray(batches, dtype=np.int32)
    else:
        sched = np.full((len(batches), batch_size), -1, np.int32)
        for i, b in enumerate(batches):
            sched[i, :len(b)] = np.asarray(b, dtype=np.int32)
    cat = np.zeros((sched.shape[0], 2 * batch_size), np.int32)
    for i, r in enumerate(sched):
        cur = r[r >= 0]
        cat[i, :len(cur)] = cur
        cat[i, len(cur):2 * len(cur)] = np.maximum(cur - 1, 0)
    return sched, torch.from_numpy(cat).to(device)


def _local_cat(sched, rank, world, device):
    """Per iteration [cur(b_loc) | max(cur-1,0)(b_loc) | pad] of THIS rank's slots (parallel.deal_slots) as one device int32 [iters, 2*bmax]."""
    from .parallel import deal_slots
    rows = [deal_slots(r, rank, world)[0] for r in sched]
    bmax = max(1, max((len(r) for r in rows), default=1))
    cat = np.zeros((len(rows), 2 * bmax), np.int32)
    for i, cur in enumerate(rows):
        cat[i, :len(cur)] = cur
        cat[i, len(cur):2 * len(cur)] = np.maximum(np.asarray(cur, np.int64) - 1, 0)
    return torch.from_numpy(cat).to(device), bmax


def exposure_align(dataset, batches, epochs, batch_size=16, lr_init=0.01, lr_final=0.001,
                   lambda_dssim=0.2, lambda_flow=0.8, iters_per_epoch=None, dist=None):
    """generate.py:354-451.  Returns (aligned images, exposure [N,3,4], losses tensor) and bakes the
    alignment into dataset.edited_images like OptDataset.exposure_align does.
    dist (parallel.Dist, world > 1): the dataset holds ALL frames on every rank; each mini-batch's slots are dealt to the ranks and the
    [N,3,4] gradient is all-reduced before every Adam step (one global exposure set, identical on all ranks)."""
    ed = dataset.edited_images
    n, _, h, w = ed.shape
    dev = ed.device
    sched, d_cat = _pack_schedule(batches, batch_size, dev)
    if iters_per_epoch is None:
        iters_per_epoch = -(-n // batch_size)      # len(DataLoader)
    expo = torch.eye(3, 4, device=dev)[None].repeat(n, 1, 1).contiguous()
    out = torch.empty_like(ed)
    L = lib()
    if dist is None or not dist.multi:
        g, m, v = (torch.zeros_like(expo) for _ in range(3))
        losses = torch.zeros(len(sched), device=dev)
        ws = torch.empty(L.tcl_stage_workspace_bytes(batch_size, h, w), dtype=torch.uint8, device=dev)
        L.tcl_exposure_align(ed, dataset.past_flows, dataset.mask_bwd, dataset.flow_shift, n, h, w, sched.ctypes.data, d_cat, len(sched),
                             iters_per_epoch, batch_size, epochs, lr_init, lr_final, lambda_dssim, lambda_flow, expo, g, m, v, losses,
                             out, ws, stream())
    else:
        from .hostlogic import expon_lr
        from .parallel import distributed_adam_loop
        lcat, bmax = _local_cat(sched, dist.rank, dist.world, dev)
        ws = torch.empty(L.tcl_stage_workspace_bytes(bmax, h, w), dtype=torch.uint8, device=dev)
        fsh = dataset.flow_shift
        total_iters = epochs * n // batch_size
        g = torch.zeros(n * 12, device=dev)

        def grad_fn(it, slots, b_glob, nvalid, p_full, g_full, loss_out):
            L.tcl_exposure_grad(ed, dataset.past_flows, dataset.mask_bwd, fsh, n, h, w, lcat[it], len(slots), b_glob, nvalid, lambda_dssim,
                                lambda_flow, p_full, g_full, loss_out, ws, stream())

        def adam_fn(it, p, gg, m, v):
            epoch, i = divmod(it, iters_per_epoch)
            lr = expon_lr(epoch * n // batch_size + i + 1, lr_init, lr_final, total_iters)
            L.tcl_adam_step(p, gg, m, v, p.numel(), lr, 0.9, 0.999, 1e-8, it + 1, stream())

        losses = distributed_adam_loop(dist, sched, expo.view(-1), g, grad_fn, adam_fn, shard_state=False)
        # Stage 2 runs REPLICATED behind this (generate.py default): every rank must hold the same exposure bits -- true when the backend's all-reduce hands
        # every rank the same sum (RCCL ring / tree do), silently false otherwise.  One 16-byte exchange says which: the ranks' bit-pattern
        # checksums must agree (max == min).  A mismatch is healed by taking rank 0's parameters, loudly.
        dist.same_bits_or_broadcast(expo, "stage-1 exposure")
        L.tcl_apply_exposure(ed, 0, expo, out, n, h, w, stream())
    dataset.edited_images = out
    return out, expo, losses


def track_ids_unique(inv, n, h, w, k, scratch=None):
    """1 when the track ids of every frame are pairwise distinct -- true of get_flowid's ids (a pixel inherits ONE id or gets a fresh one,
    utils/flow_utils.py:56-93).  Stage 2 then accumulates codebook gradients frame by frame without atomics (bit-reproducible); 0 (any other
    id layout) takes the float-atomic kernels.  One device-to-host read per run."""
    if scratch is None:
        scratch = torch.empty(k, dtype=torch.int32, device=inv.device)
    res = torch.zeros(1, dtype=torch.int32, device=inv.device)
    lib().tcl_track_ids_unique(inv, n, h, w, k, scratch, res, stream())
    return int(res.item())


def unique_tensor_optimization(dataset, unq_inv, batches, batch_size=16, feature_lr=0.05, lambda_dssim=0.2,
                               lambda_flow=0.8, lambda_tv=0.05, k=None, dist=None):
    """generate.py:453-533.  unq_inv: [N*H*W] integer tensor on the device.  Returns (images, features_dc, losses).
    dist (world > 1): ONE global codebook.  Every rank holds all frames + unq_inv and a replica of the codebook for the gather; the Adam
    state (p, m, v) is sharded by row range, the dense [3,K] gradient is reduce-scattered, updated rows are all-gathered."""
    ed = dataset.edited_images
    n, _, h, w = ed.shape
    dev = ed.device
    inv = unq_inv.to(device=dev, dtype=torch.int32).contiguous()
    if k is None:
        k = int(inv.max()) + 1
    sched, d_cat = _pack_schedule(batches, batch_size, dev)
    L = lib()
    world = dist.world if dist is not None else 1
    npad = -(-3 * k // world) * world                   # flat [3,K] padded so that the row-range shards are equal
    flat = torch.zeros(npad, device=dev)
    feat = flat[:3 * k].view(3, k)                      # channel-planar codebook (features_dc.t())
    cnt = torch.empty(k, device=dev)
    uniq = track_ids_unique(inv, n, h, w, k, scratch=cnt.view(torch.int32))
    L.tcl_scatter_mean_rgb2sh(ed, inv, feat, cnt, n, h, w, k, uniq, stream())
    del cnt
    out = torch.empty_like(ed)
    if dist is None or not dist.multi:
        g, m, v = (torch.zeros_like(feat) for _ in range(3))
        losses = torch.zeros(max(len(sched), 1), device=dev)
        ws = torch.empty(L.tcl_stage_workspace_bytes(batch_size, h, w), dtype=torch.uint8, device=dev)
        # lazy dense Adam (csrc/path2.hip): pays when the codebook is much larger than the rows one mini-batch touches (2 b frames)
        lazy = os.environ.get("TCL_ADAM_LAZY", "auto")
        use_lazy = uniq and len(sched) > 0 and (lazy == "1" or (lazy == "auto" and k > 3 * 2 * batch_size * h * w))
        lws = torch.empty(L.tcl_stage2_lazy_workspace_bytes(k, len(sched)), dtype=torch.uint8, device=dev) if use_lazy else 0
        L.tcl_unique_tensor_opt(ed, dataset.past_flows, dataset.mask_bwd, dataset.flow_shift, inv, n, h, w, k, uniq, sched.ctypes.data, d_cat,
                                len(sched), batch_size, feature_lr, lambda_dssim, lambda_flow, lambda_tv, feat, g, m, v,
                                losses, out, ws, lws, stream())
        losses = losses[:len(sched)]
    else:
        from .parallel import distributed_adam_loop
        lcat, bmax = _local_cat(sched, dist.rank, dist.world, dev)
        ws = torch.empty(L.tcl_stage_workspace_bytes(bmax, h, w), dtype=torch.uint8, device=dev)
        fsh = dataset.flow_shift
        g = torch.zeros(npad, device=dev)
        lr = feature_lr * batch_size / n                # generate.py:474

        def grad_fn(it, slots, b_glob, nvalid, p_full, g_full, loss_out):
            L.tcl_unique_tensor_grad(ed, dataset.past_flows, dataset.mask_bwd, fsh, inv, n, h, w, k, uniq, lcat[it], len(slots), b_glob, nvalid,
                                     lambda_dssim, lambda_flow, lambda_tv, p_full, g_full, loss_out, ws, stream())

        def adam_fn(it, p, gg, m, v):
            L.tcl_adam_step(p, gg, m, v, p.numel(), lr, 0.9, 0.999, 1e-15, it + 1, stream())

        losses = distributed_adam_loop(dist, sched, flat, g, grad_fn, adam_fn, shard_state=True)
        L.tcl_gather_codebook(feat, inv, 0, out, n, h, w, k, stream())
    return out, feat.t(), losses      # features_dc in the reference's [K,3] orientation (a view)
