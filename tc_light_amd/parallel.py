"""Multi-GPU choreography (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in CPU tests).

The reference is single-GPU; the sharding is this engine's design:
  * frames are split in contiguous blocks; VAE encode/decode and the xy-plane denoise touch only local frames;
  * the yt-plane pass needs every frame of a 64-frame window for each latent column: x is all-gathered once per step
    (concat_conds once per run), the (window, column-chunk) work items are dealt round-robin to the ranks, and ONE all-gather of every
    rank's OWNED (frames x columns) pieces hands each rank the noise of its own frame block (every element has one writer; round 5 -- rounds
    2-4 all-reduced a zero-filled full-size tensor: 16x the bytes on the wire);
  * stage 1/2 optimise ONE global parameter set (generate.py:472-533: one features_dc [K,3] over all frames): the decoded frames are
    all-gathered slab by slab WHILE the VAE decodes the next slab (async collectives, `gather_frames_pipelined`).  Default ("replicated",
    generate.py DEFAULTS.post_opt_mode): stage 1 deals every mini-batch's slots to the ranks (its only exchange is the 14 KB all-reduce of the
    [N,3,4] gradient) and stage 2 runs in full on every rank -- no collective; it is bound by streams that do not shrink with a rank's share
    of the mini-batch, and path 2 is bit-reproducible so the replicas agree ("replicated_all" replicates stage 1 as well: the one-GPU bits).
    "global" mode: the slots of every mini-batch are dealt to the ranks (`deal_slots`), each rank back-propagates its slots with the
    GLOBAL normalisers, and the gradients meet in a collective before the Adam step (`distributed_adam_loop`): stage 1 all-reduces the
    [N,3,4] exposure gradient (14 KB); stage 2 reduce-scatters the dense [3,K] codebook gradient, every rank owns 1/world of the
    codebook's Adam state (p, m, v: the 84 B/row/iteration stream is cut by world) and the updated rows are all-gathered.  Loss
    scalars are all-reduced once per stage.  `shard_post_opt: true` keeps round 1's collective-free approximation (each rank's frame
    block as a video of its own) as an opt-in.
All functions are backend-agnostic and are exercised with gloo / world_size 2 in tests/test_parallel_cpu.py.
"""
import torch
import torch.distributed as dist

from .hostlogic import shard_range


class Dist:
    def __init__(self, rank=0, world=1, timed=False, force_collectives=False):
        self.rank, self.world = rank, world
        # force_collectives: issue every collective even at world == 1 (tests/test_gpu_rccl.py: the one GPU of a test box still runs each call through
        # the RCCL backend -- communicator setup, the device-side kernels, stream ordering -- where the world == 1 shortcuts below would skip it)
        self.multi = world > 1 or bool(force_collectives)
        # timed=True (bench.py, N > 1): every collective is bracketed by two EVENTS on the stream it is issued from -- resolved by
        # collect_stats() after the pass, so the timed region has no device-wide synchronise in it and compute / communication overlap as in an
        # un-instrumented run (rounds 3-4 synchronised the device on both sides of every call: ADVICE r4) -- and its bytes are counted.  The
        # event interval holds the collective AND the wait for the slowest rank to reach it.
        self.timed = timed
        self.stats = {}
        self._events = []

    def _coll(self, name, nbytes, fn):
        if not self.timed or not torch.cuda.is_available():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self._events.append((name, int(nbytes), e0, e1))
        return out

    def collect_stats(self):
        """Resolve the event pairs recorded so far (synchronises on them) -> {name: {calls, seconds, bytes}}, cumulative since reset_stats()."""
        for name, nbytes, e0, e1 in self._events:
            e1.synchronize()
            s = self.stats.setdefault(name, {"calls": 0, "seconds": 0.0, "bytes": 0})
            s["calls"] += 1; s["seconds"] += e0.elapsed_time(e1) * 1e-3; s["bytes"] += nbytes
        self._events = []
        return self.stats

    def reset_stats(self):
        self.stats = {}
        self._events = []

    @classmethod
    def from_env(cls):
        if dist.is_available() and dist.is_initialized():
            return cls(dist.get_rank(), dist.get_world_size())
        return cls()

    def range(self, n):
        return shard_range(n, self.rank, self.world)

    def gather_frames(self, x_local, n_total):
        """all-gather along dim 0 of uneven contiguous shards -> [n_total, ...]."""
        if not self.multi:
            return x_local
        nmax = -(-n_total // self.world)
        pad = torch.zeros((nmax,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        pad[:x_local.shape[0]] = x_local
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self._coll("all_gather_frames", pad.numel() * pad.element_size() * self.world, lambda: dist.all_gather(out, pad))
        parts = []
        for r in range(self.world):
            lo, hi = shard_range(n_total, r, self.world)
            parts.append(out[r][:hi - lo])
        return torch.cat(parts)

    def gather_frames_pipelined(self, produce, n_local, n_total, slab=8):
        """gather_frames of frames that are still being produced: produce(a, b) -> this rank's frames [a, b) (e.g. the VAE decode of a slab of
        latents).  Every slab is handed to an ASYNC all-gather as soon as it exists, so the transfer of slab s rides under the production of slab
        s + 1 (RCCL runs collectives on its own stream; the only exposed transfer is the last slab's).  Same result as
        gather_frames(produce(0, n_local), n_total)."""
        if not self.multi:
            return produce(0, n_local)
        if n_local < 1:
            raise ValueError(f"gather_frames_pipelined: rank {self.rank} holds no frame ({n_total} frames over {self.world} ranks): every rank needs at least one")
        nmax = -(-n_total // self.world)
        nslab = -(-nmax // slab)
        works, outs, shape = [], [], None
        for s_ in range(nslab):
            a, b = min(s_ * slab, n_local), min((s_ + 1) * slab, n_local)
            part = produce(a, b) if b > a else None
            if shape is None:
                shape, dt, dv = tuple(part.shape[1:]), part.dtype, part.device
            pad = torch.zeros((slab,) + shape, dtype=dt, device=dv)
            if part is not None:
                pad[:b - a] = part
            out = torch.empty((self.world * slab,) + shape, dtype=dt, device=dv)
            works.append(dist.all_gather_into_tensor(out, pad, async_op=True))
            outs.append(out)
            if self.timed:
                st = self.stats.setdefault("all_gather_decoded_async", {"calls": 0, "seconds": 0.0, "bytes": 0})
                st["calls"] += 1; st["bytes"] += out.numel() * out.element_size()
        self._coll("all_gather_decoded_async_wait", 0, lambda: [w.wait() for w in works])
        parts = []
        for r in range(self.world):
            lo, hi = shard_range(n_total, r, self.world)
            for s_ in range(nslab):
                cnt = min((s_ + 1) * slab, hi - lo) - s_ * slab
                if cnt > 0:
                    parts.append(outs[s_][r * slab:r * slab + cnt])
        return torch.cat(parts)

    def my_items(self, items):
        """Round-robin deal of yt-plane work items (identical list on every rank)."""
        return items[self.rank::self.world]

    def reduce_full(self, full):
        """Sum the per-rank partially filled full-size tensors (disjoint support) in place.  (Rounds 2-4's yt exchange; kept for A/B:
        TCL_YT_EXCHANGE=allreduce.)"""
        if self.multi:
            self._coll("all_reduce_yt_noise", full.numel() * full.element_size(), lambda: dist.all_reduce(full, op=dist.ReduceOp.SUM))
        return full

    def all_gather_flat(self, name, send):
        """-> [world, len(send)]: every rank's equally long flat buffer."""
        recv = torch.empty((self.world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        self._coll(name, recv.numel() * recv.element_size(), lambda: dist.all_gather_into_tensor(recv.view(-1), send))
        return recv

    def all_reduce_sum(self, t):
        if self.multi:
            self._coll("all_reduce", t.numel() * t.element_size(), lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        return t

    def reduce_scatter_sum(self, full, out):
        """out[i] = sum over ranks of full[rank*len(out) + i]  (full.numel() == world * out.numel()).  RCCL: one reduce_scatter; gloo has
        no reduce_scatter, so the CPU tests take the all_reduce + slice route (same values)."""
        if not self.multi:
            out.copy_(full)
        elif dist.get_backend() == "nccl":
            self._coll("reduce_scatter", full.numel() * full.element_size(), lambda: dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM))
        else:
            self._coll("reduce_scatter", full.numel() * full.element_size(), lambda: dist.all_reduce(full, op=dist.ReduceOp.SUM))
            n = out.numel()
            out.copy_(full[self.rank * n:(self.rank + 1) * n])
        return out

    def all_gather_into(self, full, shard):
        """full = concatenation over ranks of shard (equal sizes)."""
        if not self.multi:
            full.copy_(shard)
        else:
            self._coll("all_gather_rows", full.numel() * full.element_size(), lambda: dist.all_gather_into_tensor(full, shard))
        return full

    def same_bits_or_broadcast(self, t, what="tensor"):
        """All ranks should hold the same bits of `t` (a replicated parameter after an all-reduced update).  Checks it with one small all-reduce (MAX of
        [c, -c], c = a 53-bit checksum of the bit pattern held in an f64) and, on a mismatch, warns and broadcasts rank 0's copy."""
        if not self.multi:
            return True
        v = t.detach().contiguous().view(torch.int32).reshape(-1).to(torch.int64)
        c = ((v * (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 8191 + 1)).sum() % (1 << 52)).to(torch.float64)
        pair = torch.stack([c, -c])
        dist.all_reduce(pair, op=dist.ReduceOp.MAX)
        same = bool((pair[0] == -pair[1]).item())
        if not same:
            import warnings
            warnings.warn(f"{what}: the ranks' copies differ after the all-reduced update (backend {dist.get_backend()}): taking rank 0's")
            dist.broadcast(t, src=0)
        return same

    def barrier(self):
        if self.multi:
            dist.barrier()

    def max_float(self, v, device):
        if not self.multi:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def yt_pieces(items, rank, world):
    """The (frame range, columns) pieces of the yt noise that `rank` produces: one per frame window -- the columns of all its items of that
    window (item = (window start, window length, columns, scale_upto, nkeep) writes frames [start, start + nkeep) x columns, and every
    (frame, column) of the clip has exactly one writer item: generate.py:265-278)."""
    pieces = []
    for it in items[rank::world]:
        f0, f1 = it[0], it[0] + it[4]
        if pieces and pieces[-1][:2] == (f0, f1):
            pieces[-1][2].extend(int(c) for c in it[2])
        else:
            pieces.append((f0, f1, [int(c) for c in it[2]]))
    return pieces


def sharded_temporal_pass(d, x_local, cc_full, n_total, items, compute, x_full=None):
    """yt-plane pass.  items: list of (window_start, window_len, column_chunk, scale_upto, nkeep) identical on every rank;
    compute(x_full, cc_full, items, noises_t_full) takes this rank's items of ONE window (same window length, reference order) and
    writes their columns for the window's frames into noises_t_full.  Returns this rank's frame block of the assembled noises_t.
    Exchange: every rank packs the (frames x columns) pieces it produced into one flat buffer, ONE all-gather moves them, and every rank
    copies the parts that fall into its own frame block out of it -- 1/world of the clip's noise per rank on the wire."""
    import os
    if x_full is None:
        x_full = d.gather_frames(x_local, n_total)
    nt_full = torch.zeros_like(x_full)
    mine, group = d.my_items(items), []
    for it in mine + [None]:                       # consecutive items of one window go to the UNet together
        if group and (it is None or it[:2] != group[0][:2]):
            compute(x_full, cc_full, group, nt_full)
            group = []
        if it is not None:
            group.append(it)
    lo, hi = d.range(n_total)
    if not d.multi:
        return nt_full[lo:hi]
    if os.environ.get("TCL_YT_EXCHANGE", "allgather") == "allreduce":
        d.reduce_full(nt_full)
        return nt_full[lo:hi]
    _, C, h, w = nt_full.shape
    unit = C * h
    # the piece lists and their column-index tensors depend only on the step's item list: built once per distinct list and kept (world x windows
    # blocking host-to-device copies per denoising step sat inside the timed loop; the yt chunk draws differ per step, so the cache holds a few entries)
    cache = d.__dict__.setdefault("_yt_cache", {})
    key = (d.world, tuple((it[0], it[4], tuple(int(c) for c in it[2])) for it in items), str(nt_full.device))
    hit = cache.get(key)
    if hit is None:
        if len(cache) > 64:
            cache.clear()
        plists = [yt_pieces(items, r, d.world) for r in range(d.world)]
        idx = {}
        for pl in plists:
            for _, _, cols in pl:
                idx.setdefault(tuple(cols), torch.tensor(cols, dtype=torch.int64, device=nt_full.device))
        hit = cache[key] = (plists, idx)
    plists, idx = hit
    pmax = max(sum((f1 - f0) * len(cols) for f0, f1, cols in pl) for pl in plists) * unit
    dev = nt_full.device
    cidx = lambda cols: idx[tuple(cols)]
    send = torch.zeros(pmax, dtype=nt_full.dtype, device=dev)
    off = 0
    for f0, f1, cols in plists[d.rank]:
        blk = nt_full[f0:f1].index_select(3, cidx(cols))           # [frames, C, h, columns]
        send[off:off + blk.numel()] = blk.reshape(-1)
        off += blk.numel()
    recv = d.all_gather_flat("all_gather_yt_noise", send)
    out = torch.zeros((hi - lo, C, h, w), dtype=nt_full.dtype, device=dev)
    for r, pl in enumerate(plists):
        off = 0
        for f0, f1, cols in pl:
            n = (f1 - f0) * unit * len(cols)
            a, b = max(f0, lo), min(f1, hi)
            if a < b:
                out[a - lo:b - lo].index_copy_(3, cidx(cols), recv[r, off:off + n].view(f1 - f0, C, h, len(cols))[a - f0:b - f0])
            off += n
    return out


def deal_slots(row, rank, world):
    """One mini-batch (frame ids, -1 = padding of a short batch) -> (this rank's slots, slots of the whole batch, how many of those have
    id > 0).  Round-robin, like the yt items: every rank gets floor/ceil(b/world) slots."""
    cur = [int(f) for f in row if int(f) >= 0]
    return cur[rank::world], len(cur), sum(f > 0 for f in cur)


def distributed_adam_loop(d, sched, p_full, g_full, grad_fn, adam_fn, shard_state, new_zeros=None):
    """The stage-1 / stage-2 optimisation loop (generate.py:392-433, :492-523) over `sched` ([iters][batch] frame ids, -1 padded) with the
    mini-batch split over the ranks and ONE parameter set.

    p_full: flat parameter tensor, identical on every rank (numel divisible by world when shard_state); g_full: flat gradient
    accumulator, zero on entry.  grad_fn(it, slots, b_glob, nvalid_glob, p_full, g_full, loss_out) adds this rank's partial gradient
    (global normalisers) into g_full and writes its share of the loss into loss_out ([1] view); it is skipped for a rank without slots.
    adam_fn(it, p, g, m, v) is one Adam step on (a shard of) the parameters and leaves g zero.
    shard_state=False: all_reduce(g) and a replicated step (bit-identical on all ranks: the all-reduced gradient is).  shard_state=True:
    reduce_scatter(g) -> Adam on the rank's 1/world of (p, m, v) -> all_gather of the updated rows into p_full.
    Returns the per-iteration losses (all-reduced once at the end)."""
    new_zeros = new_zeros or (lambda n: torch.zeros(n, dtype=p_full.dtype, device=p_full.device))
    n = p_full.numel()
    losses = new_zeros(max(len(sched), 1))
    if shard_state and d.multi:
        assert n % d.world == 0, "pad the flat parameter to a multiple of the world size"
        sz = n // d.world
        p = p_full[d.rank * sz:(d.rank + 1) * sz].clone()
        g_sh, m, v = new_zeros(sz), new_zeros(sz), new_zeros(sz)
    else:
        shard_state = False
        m, v = new_zeros(n), new_zeros(n)
    for it, row in enumerate(sched):
        slots, b_glob, nvalid = deal_slots(row, d.rank, d.world)
        if slots:
            grad_fn(it, slots, b_glob, nvalid, p_full, g_full, losses[it:it + 1])
        if shard_state:
            d.reduce_scatter_sum(g_full, g_sh)
            g_full.zero_()
            adam_fn(it, p, g_sh, m, v)
            d.all_gather_into(p_full, p)
        else:
            d.all_reduce_sum(g_full)
            adam_fn(it, p_full, g_full, m, v)
    d.all_reduce_sum(losses)
    return losses[:len(sched)]
