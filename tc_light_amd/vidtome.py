"""VidToMe token merging, host side (device-resident state, HIP kernels for all arithmetic).

Mirrors utils/VidToMe/vidtome/patch.py: `apply_patch(...)` arguments live in `VidToMe.args`, `update_patch(pipe,
global_tokens=None)` is `VidToMe.reset_global_tokens()`, and `compute_merge` (patch.py:14-91) runs per patched
transformer block.  Differences, all deliberate and documented in DESIGN.md:
  * the global-token bank of each block stays in HBM (the reference copies it to the CPU and back, patch.py:65-82);
  * the random choices (`randf`, merge.py:56-58; the src/dst coin, patch.py:61) come from an explicit host stream of
    draws shared by all blocks of one UNet forward -- in the reference every block owns a generator forked from the same
    state, so they draw identical numbers in lock-step (vidtome/utils.py:18-30, patch.py:215-231);
  * ties in the greedy matching are broken deterministically (csrc/merge.hip header).
"""
import math

import numpy as np
import torch

from .lib import lib, stream

H16 = torch.float16
I32 = torch.int32


class VidToMe:
    def __init__(self, device, local_merge_ratio=0.6, merge_global=True, global_merge_ratio=0.5, max_downsample=2, seed=123,
                 batch_size=2, align_batch=True, target_stride=4, global_rand=0.5, enabled=True):
        if batch_size != 2 or not align_batch:
            raise NotImplementedError("TC-Light runs VidToMe with batch_size=2 (uncond, cond) and align_batch=True")
        self.dev = torch.device(device)
        self.args = dict(local_merge_ratio=local_merge_ratio, merge_global=merge_global, global_merge_ratio=global_merge_ratio,
                         max_downsample=max_downsample, seed=seed, batch_size=batch_size, align_batch=align_batch,
                         target_stride=target_stride, global_rand=global_rand)
        self.enabled = enabled
        self.rng = np.random.default_rng(seed)
        self.banks = {}                 # block name -> [2, Tb, C] f16 (module.global_tokens, patch.py:60-82)
        self._pos = {}
        self._ws = None
        self.draws = None               # optional injected (randf, coin) for the next forwards (parity tests)
        self.trace = None               # set to a list to record per-block maps (parity tests)
        self.L = lib()

    # ---- reference surface
    def reset_global_tokens(self):      # vidtome.update_patch(pipe, global_tokens=None)  (generate_utils.py:235-238)
        self.banks.clear()

    def begin_forward(self, F, size):
        """One UNet forward over one chunk: fix the draws every patched block will see (lock-step generators of the reference)."""
        self.begin_step([F], size)
        self.select_chunk(0)

    def begin_step(self, Fs, size):
        """One UNet pass over the chunks Fs (reference chunk order): draw each chunk's (randf, coin) in that order."""
        self.size = size
        self._chunks = []
        for F in Fs:
            if self.draws is not None:
                randf, coin = self.draws.pop(0)
            else:
                ts = min(self.args["target_stride"], F)
                randf = int(self.rng.integers(0, ts)) if F > 1 else -1
                coin = float(self.rng.random())
            self._chunks.append((F, randf, coin))

    def select_chunk(self, i):
        self.F, self.randf, self.coin = self._chunks[i]

    def end_forward(self):
        pass

    # ---- helpers
    def _positions(self, F, N, randf):
        key = (F, N, randf)
        hit = self._pos.get(key)
        if hit is None:
            idx = torch.arange(F * N, dtype=I32)
            dst = (idx // N) % min(self.args["target_stride"], F) == randf     # merge.py:59-60 (unm_pre = 0)
            hit = self._pos[key] = (idx[~dst].contiguous().to(self.dev), idx[dst].contiguous().to(self.dev))
        return hit

    def _range(self, lo, hi):
        key = ("r", lo, hi)
        hit = self._pos.get(key)
        if hit is None:
            hit = self._pos[key] = torch.arange(lo, hi, dtype=I32, device=self.dev)
        return hit

    def _match(self, tokens, T, C, a_pos, na, b_pos, nb, ratio, tbs=None, affine=None):
        """tokens: two [T, C] slices tbs elements apart (default T*C: a contiguous [2, T, C]) -> (mrg [na-r+nb], unm [T]) int32 maps.
        affine = (a_split, a_gap, b0): a_pos[i] = i if i < a_split else i + a_gap, b_pos[j] = b0 + j (true of every VidToMe match; lets the
        C = 320 matches take the strip-resident kernel, csrc/merge.hip::k_tome_match320 -- same maps, bit for bit)."""
        L = self.L
        r = min(na, int(na * ratio))                                            # merge.py:90
        metric = torch.empty(2 * T, C, dtype=H16, device=self.dev)
        if tbs is None or tbs == T * C:
            L.tcl_tome_normalize_f16(tokens, metric, 2 * T, C, stream())
        else:
            flat = tokens.reshape(-1)
            L.tcl_tome_normalize_f16(flat, metric, T, C, stream())
            L.tcl_tome_normalize_f16(flat[tbs:], metric[T:], T, C, stream())
        need = L.tcl_tome_match_workspace_bytes(na)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.dev)        # zeroed once; every match leaves it zero again
        mrg = torch.empty(na - r + nb, dtype=I32, device=self.dev)
        unm = torch.empty(T, dtype=I32, device=self.dev)
        if affine is not None:
            L.tcl_tome_match_affine_f16(metric, T * C, 2, C, a_pos, na, b_pos, nb, r, affine[0], affine[1], affine[2], mrg, unm, self._ws, stream())
        else:
            L.tcl_tome_match_f16(metric, T * C, 2, C, a_pos, na, b_pos, nb, r, mrg, unm, self._ws, stream())
        return mrg, unm, na - r + nb

    # ---- patch.py:14-91
    def merges(self, N):
        """patch.py:15-18: does a block with N tokens per frame merge at all?"""
        if not self.enabled:
            return False
        return int(math.ceil(math.sqrt((self.size[0] * self.size[1]) // N))) <= self.args["max_downsample"]

    def compute_merge(self, name, x, F, N, C, _unused=None, xbs=None):
        """x: norm1 output of one chunk: the unconditional [F*N, C] rows at x, the conditional ones xbs elements further (default
        F*N*C: a contiguous [2F, N, C] == joined [2, F*N, C]).  Returns None when this block is not merged, else
        (merged [2,T,C], unm int32 [F*N] or None for identity, T)."""
        a = self.args
        if not self.merges(N):
            return None
        L = self.L
        if xbs is None:
            xbs = F * N * C
        if F > a["target_stride"]:
            # patch.py:44-56 merges longer chunks in several randframe rounds (8 -> 2 -> 1) carrying the unmerged tokens along; TC-Light
            # never configures chunk_size > target_stride (4), so only the single round is built -- refuse instead of mis-indexing.
            raise NotImplementedError(f"VidToMe local merging of {F}-frame chunks: only chunks of <= target_stride "
                                      f"({a['target_stride']}) frames (one randframe round) are implemented")
        if F > 1:
            a_pos, b_pos = self._positions(F, N, self.randf)
            mrg1, unm1, TL = self._match(x, F * N, C, a_pos, a_pos.numel(), b_pos, b_pos.numel(), a["local_merge_ratio"], tbs=xbs,
                                         affine=(self.randf * N, N, self.randf * N))      # dst = the N tokens of frame randf, src = the rest
            local = torch.empty(2, TL, C, dtype=H16, device=self.dev)
            L.tcl_gather_rows_f16(x, xbs, 0, 0, mrg1, local, TL * C, 2, TL, C, stream())
        else:
            mrg1 = unm1 = None
            TL = N                                       # F == 1: nothing to merge locally; keep the [2, T, C] layout
            if xbs == N * C:
                local = x.reshape(-1)[:2 * N * C].view(2, N, C)
            else:
                local = torch.empty(2, N, C, dtype=H16, device=self.dev)
                L.tcl_gather_rows_f16(x, xbs, 0, 0, 0, local, N * C, 2, N, C, stream())
        if not a["merge_global"]:
            return local, unm1, TL
        bank = self.banks.get(name)
        if bank is None:                                                        # patch.py:81-82: the first chunk seeds the bank
            self.banks[name] = local if F > 1 else local.clone()
            if self.trace is not None:
                self.trace.append(dict(name=name, F=F, unm=unm1, gather=mrg1, mrg1=mrg1, T=TL))
            return local, unm1, TL
        Tb = bank.shape[1]
        if self.coin > a["global_rand"]:                                        # patch.py:61-65: local tokens are src
            src_len, loff, boff = TL, 0, TL
        else:                                                                   # patch.py:66-70: bank tokens are src
            src_len, loff, boff = Tb, Tb, 0
        T = TL + Tb
        cat = torch.empty(2, T, C, dtype=H16, device=self.dev)
        L.tcl_gather_rows_f16(local, TL * C, 0, 0, 0, cat[:, loff:], T * C, 2, TL, C, stream())
        L.tcl_gather_rows_f16(bank, Tb * C, 0, 0, 0, cat[:, boff:], T * C, 2, Tb, C, stream())
        mrg2, unm2, Tm = self._match(cat, T, C, self._range(0, src_len), src_len, self._range(src_len, T), T - src_len,
                                     a["global_merge_ratio"], affine=(src_len, 0, src_len))
        merged = torch.empty(2, Tm, C, dtype=H16, device=self.dev)
        L.tcl_gather_rows_f16(cat, T * C, 0, 0, mrg2, merged, Tm * C, 2, Tm, C, stream())
        unm = torch.empty(F * N, dtype=I32, device=self.dev)
        L.tcl_index_compose(unm2, unm1 if unm1 is not None else 0, loff, F * N, unm, stream())    # 2s-unmerge then randframe-unmerge
        bmap = torch.empty(TL, dtype=I32, device=self.dev)                     # bank <- u(merged_tokens) (patch.py:80)
        L.tcl_index_compose(mrg2, unm2[loff:], 0, TL, bmap, stream())
        nb_ = torch.empty(2, TL, C, dtype=H16, device=self.dev)
        L.tcl_gather_rows_f16(cat, T * C, 0, 0, bmap, nb_, TL * C, 2, TL, C, stream())
        self.banks[name] = nb_
        if self.trace is not None:
            self.trace.append(dict(name=name, F=F, unm=unm, mrg2=mrg2, mrg1=mrg1, T=Tm, loff=loff, boff=boff, TL=TL, bmap=bmap))
        return merged, unm, Tm
