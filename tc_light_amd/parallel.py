"""Multi-GPU choreography (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in CPU tests).

The reference is single-GPU; the sharding is this engine's design:
  * frames are split in contiguous blocks; VAE encode/decode and the xy-plane denoise touch only local frames;
  * the yt-plane pass needs every frame of a 64-frame window for each latent column: x is all-gathered once per step
    (concat_conds once per run), the (window, column-chunk) work items are dealt round-robin to the ranks, each rank writes
    its columns into a zero full-size noise tensor and ONE all-reduce(SUM) assembles it (every element has one writer);
  * stage 1/2 optimise ONE global parameter set (generate.py:472-533: one features_dc [K,3] over all frames): the decoded frames are
    all-gathered once.  Default ("replicated", generate.py DEFAULTS.post_opt_mode): every rank then runs the whole optimisation itself --
    no collective; both stages are bound by streams that do not shrink with a rank's share of the mini-batch, and path 2 is bit-reproducible
    so the replicas agree.  "global" mode: the slots of every mini-batch are dealt to the ranks (`deal_slots`), each rank back-propagates its slots with the
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
    def __init__(self, rank=0, world=1, timed=False):
        self.rank, self.world = rank, world
        # timed=True (bench.py, N > 1): every collective is bracketed by a device synchronise and the host clock, so a scaling run can say
        # how long the ranks sat in collectives (waiting for the slowest rank included) and how many bytes each one moved.  ~3 per denoising step.
        self.timed = timed
        self.stats = {}

    def _coll(self, name, nbytes, fn):
        if not self.timed:
            return fn()
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        s = self.stats.setdefault(name, {"calls": 0, "seconds": 0.0, "bytes": 0})
        s["calls"] += 1; s["seconds"] += time.perf_counter() - t0; s["bytes"] += int(nbytes)
        return out

    def reset_stats(self):
        self.stats = {}

    @classmethod
    def from_env(cls):
        if dist.is_available() and dist.is_initialized():
            return cls(dist.get_rank(), dist.get_world_size())
        return cls()

    def range(self, n):
        return shard_range(n, self.rank, self.world)

    def gather_frames(self, x_local, n_total):
        """all-gather along dim 0 of uneven contiguous shards -> [n_total, ...]."""
        if self.world == 1:
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

    def my_items(self, items):
        """Round-robin deal of yt-plane work items (identical list on every rank)."""
        return items[self.rank::self.world]

    def reduce_full(self, full):
        """Sum the per-rank partially filled full-size tensors (disjoint support) in place."""
        if self.world > 1:
            self._coll("all_reduce_yt_noise", full.numel() * full.element_size(), lambda: dist.all_reduce(full, op=dist.ReduceOp.SUM))
        return full

    def all_reduce_sum(self, t):
        if self.world > 1:
            self._coll("all_reduce", t.numel() * t.element_size(), lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        return t

    def reduce_scatter_sum(self, full, out):
        """out[i] = sum over ranks of full[rank*len(out) + i]  (full.numel() == world * out.numel()).  RCCL: one reduce_scatter; gloo has
        no reduce_scatter, so the CPU tests take the all_reduce + slice route (same values)."""
        if self.world == 1:
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
        if self.world == 1:
            full.copy_(shard)
        else:
            self._coll("all_gather_rows", full.numel() * full.element_size(), lambda: dist.all_gather_into_tensor(full, shard))
        return full

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def max_float(self, v, device):
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def sharded_temporal_pass(d, x_local, cc_full, n_total, items, compute, x_full=None):
    """yt-plane pass.  items: list of (window_start, window_len, column_chunk, scale_upto, nkeep) identical on every rank;
    compute(x_full, cc_full, items, noises_t_full) takes this rank's items of ONE window (same window length, reference order) and
    writes their columns for the window's frames into noises_t_full.  Returns this rank's frame block of the assembled noises_t."""
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
    d.reduce_full(nt_full)
    lo, hi = d.range(n_total)
    return nt_full[lo:hi]


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
    if shard_state and d.world > 1:
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
