"""Multi-GPU choreography (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in CPU tests).

The reference is single-GPU; the sharding is this engine's design:
  * frames are split in contiguous blocks; VAE encode/decode and the xy-plane denoise touch only local frames;
  * the yt-plane pass needs every frame of a 64-frame window for each latent column: x is all-gathered once per step
    (concat_conds once per run), the (window, column-chunk) work items are dealt round-robin to the ranks, each rank writes
    its columns into a zero full-size noise tensor and ONE all-reduce(SUM) assembles it (every element has one writer);
  * stage 1/2 run replicated on the all-gathered decoded frames (their codebook gradient all-reduce would cost more over
    xGMI than the whole stage).
All functions are backend-agnostic and are exercised with gloo / world_size 2 in tests/test_parallel_cpu.py.
"""
import torch
import torch.distributed as dist

from .hostlogic import shard_range


class Dist:
    def __init__(self, rank=0, world=1):
        self.rank, self.world = rank, world

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
        dist.all_gather(out, pad)
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
            dist.all_reduce(full, op=dist.ReduceOp.SUM)
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
