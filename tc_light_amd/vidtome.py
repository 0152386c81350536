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
import os

import numpy as np
import torch

from .lib import lib, stream

H16 = torch.float16
I32 = torch.int32


class VidToMe:
    def __init__(self, device, local_merge_ratio=0.6, merge_global=True, global_merge_ratio=0.5, max_downsample=2, seed=123,
                 batch_size=2, align_batch=True, target_stride=4, global_rand=0.5, enabled=True):
        if batch_size != 2:
            raise NotImplementedError("TC-Light runs VidToMe with batch_size=2 (uncond, cond)")
        self.dev = torch.device(device)
        self.args = dict(local_merge_ratio=local_merge_ratio, merge_global=merge_global, global_merge_ratio=global_merge_ratio,
                         max_downsample=max_downsample, seed=seed, batch_size=batch_size, align_batch=align_batch,
                         target_stride=target_stride, global_rand=global_rand)
        self.enabled = enabled
        self.rng = np.random.default_rng(seed)
        self.banks = {}                 # block name -> [2, Tb, C] f16 (module.global_tokens, patch.py:60-82); may be a strided view (see _home)
        self._bank_met = {}             # block name -> the bank's cosine-normalised rows, same shape / strides (None: not carried, normalise on use)
        self._home = {}                 # block name -> the [src | dst] blocks of the NEXT chunk, which already hold this bank in their slot (compute_merge)
        self._cur = (None, 0)
        self._pos = {}
        self._ws = None
        self.draws = None               # optional injected (randf, coin) for the next forwards (parity tests); randf: int or one per round
        self.trace = None               # set to a list to record per-block maps (parity tests)
        self.L = lib()

    # ---- reference surface
    def reset_global_tokens(self):      # vidtome.update_patch(pipe, global_tokens=None)  (generate_utils.py:235-238)
        self.banks.clear()
        self._bank_met.clear()
        self._home.clear()

    def begin_forward(self, F, size):
        """One UNet forward over one chunk: fix the draws every patched block will see (lock-step generators of the reference)."""
        self.begin_step([F], size)
        self.select_chunk(0)

    def round_frames(self, F, randfs=None):
        """Frame counts of the randframe rounds of an F-frame chunk (patch.py:43-56): the dst set of a round is every frame f with
        f % min(target_stride, frames) == randf, and those frames are what the next round sees -- 4 -> [4], 8 -> [8, 2], 16 -> [16, 4].
        Draws one randf per round (or checks the given ones) -> (frame counts, randfs)."""
        counts, out, cur, k = [], [], F, 0
        while cur > 1:
            ts = min(self.args["target_stride"], cur)
            rf = int(randfs[k]) if randfs is not None else int(self.rng.integers(0, ts))
            counts.append(cur); out.append(rf)
            cur = sum(1 for f in range(cur) if f % ts == rf)
            k += 1
        return counts, out

    def begin_step(self, Fs, size):
        """One UNet pass over the chunks Fs (reference chunk order): draw each chunk's (randf per round, coin) in that order."""
        self.size = size
        self._chunks = []
        for F in Fs:
            if self.draws is not None:
                randf, coin = self.draws.pop(0)
                randfs = self.round_frames(F, list(randf) if isinstance(randf, (list, tuple)) else [randf])[1]
            else:
                randfs = self.round_frames(F)[1]
                coin = float(self.rng.random())
            self._chunks.append((F, randfs, coin))
        return self._chunks

    def select_chunk(self, i, chunks=None):
        """chunks: a list begin_step returned earlier (two groups of chunks in flight: unet.py forward_pair); default the last begin_step's."""
        lst = chunks if chunks is not None else self._chunks
        self._cur = (lst, i)
        self.F, self.randfs, self.coin = lst[i]
        self.randf = self.randfs[0] if self.randfs else -1

    def end_forward(self):
        pass

    # ---- helpers
    def _positions(self, F, N, randf, unm_pre=0):
        """merge.py:44-67: the sequence is [unm_pre | F frames of N tokens]; src = the tokens of the frames f with f % min(stride, F) != randf,
        dst = the other frames' tokens followed by the unm_pre leading tokens."""
        key = (F, N, randf, unm_pre)
        hit = self._pos.get(key)
        if hit is None:
            idx = torch.arange(F * N, dtype=I32)
            dst = (idx // N) % min(self.args["target_stride"], F) == randf
            b = torch.cat([idx[dst] + unm_pre, torch.arange(unm_pre, dtype=I32)])
            hit = self._pos[key] = ((idx[~dst] + unm_pre).contiguous().to(self.dev), b.contiguous().to(self.dev))
        return hit

    def _range(self, lo, hi):
        key = ("r", lo, hi)
        hit = self._pos.get(key)
        if hit is None:
            hit = self._pos[key] = torch.arange(lo, hi, dtype=I32, device=self.dev)
        return hit

    # Maps are int32 tensors: 1-D when the two batch entries share the matching (align_batch, merge.py:93-108), [2, n] when every entry has
    # its own (merge.py:109-118).  The three helpers below hide the difference from compute_merge.
    def _gather(self, s1, bs1, mp, out, bso, n, C, nb=2):
        L = self.L
        if mp is None or mp.dim() == 1:
            L.tcl_gather_rows_f16(s1, bs1, 0, 0, mp if mp is not None else 0, out, bso, nb, n, C, stream())
        else:
            f1, fo = s1.reshape(-1), out.reshape(-1)
            for b in range(nb):
                L.tcl_gather_rows_f16(f1[b * bs1:], bs1, 0, 0, mp[b], fo[b * bso:], bso, 1, n, C, stream())

    def _compose(self, outer, inner, off, n):
        """out[i] = outer[off + inner[i]] (inner None = identity), per batch entry when either map is."""
        L = self.L
        per = outer.dim() == 2 or (inner is not None and inner.dim() == 2)
        if not per:
            out = torch.empty(n, dtype=I32, device=self.dev)
            L.tcl_index_compose(outer, inner if inner is not None else 0, off, n, out, stream())
            return out
        nb = outer.shape[0] if outer.dim() == 2 else inner.shape[0]
        out = torch.empty(nb, n, dtype=I32, device=self.dev)
        for b in range(nb):
            L.tcl_index_compose(outer[b] if outer.dim() == 2 else outer, (inner[b] if inner.dim() == 2 else inner) if inner is not None else 0,
                                off, n, out[b], stream())
        return out

    def unmerge_add(self, h, bsh, y, T, unm, n, C, nb=2):
        """h[b][i] += y[b][unm[i]]: unmerge of attn1's output + residual (patch.py:178-179); unm None = identity."""
        L = self.L
        if unm is None or unm.dim() == 1:
            L.tcl_gather_add_rows_f16(h, bsh, y, T * C, unm if unm is not None else 0, nb, n, C, stream())
        else:
            fh, fy = h.reshape(-1), y.reshape(-1)
            for b in range(nb):
                L.tcl_gather_add_rows_f16(fh[b * bsh:], bsh, fy[b * T * C:], T * C, unm[b], 1, n, C, stream())

    def _match(self, tokens, T, C, a_pos, na, b_pos, nb, ratio, tbs=None, affine=None, metric=None, ne=2):
        """tokens: two [T, C] slices tbs elements apart (default T*C: a contiguous [2, T, C]) -> (mrg [na-r+nb], unm [T]) int32 maps ([2, .]
        each without align_batch).  metric: the tokens' cosine-normalised rows in the same layout, when the producer already wrote them
        (tcl_layernorm_metric_f16); otherwise they are normalised here.  ne: batch entries present (1: the unconditional / conditional pair is
        known to be identical, see compute_merge).
        affine = (a_split, a_gap, b0): a_pos[i] = i if i < a_split else i + a_gap, b_pos[j] = b0 + j (true of the single-round VidToMe matches;
        lets the C = 320 matches take the strip-resident kernel, csrc/merge.hip::k_tome_match320 -- same maps, bit for bit)."""
        L = self.L
        r = min(na, int(na * ratio))                                            # merge.py:90
        mbs = T * C                                                             # elements between the two batch entries of the metric
        if metric is not None:
            metric, mbs = metric.reshape(-1), (T * C if tbs is None else tbs)
        else:
            metric = torch.empty(ne * T * C, dtype=H16, device=self.dev)
            if ne == 1 or tbs is None or tbs == T * C:
                L.tcl_tome_normalize_f16(tokens, metric, ne * T, C, stream())
            else:
                flat = tokens.reshape(-1)
                L.tcl_tome_normalize_f16(flat, metric, T, C, stream())
                L.tcl_tome_normalize_f16(flat[tbs:], metric[T * C:], T, C, stream())
        need = L.tcl_tome_match_workspace_bytes(na)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.dev)        # zeroed once; every match leaves it zero again
        aligned = self.args["align_batch"]
        shape = (na - r + nb,) if aligned else (ne, na - r + nb)
        mrg = torch.empty(shape, dtype=I32, device=self.dev)
        unm = torch.empty((T,) if aligned else (ne, T), dtype=I32, device=self.dev)
        for b in range(1 if aligned else ne):                                   # aligned: ONE matching over both entries (scores concatenated along dst)
            mt, mo, uo, Bt = (metric, mrg, unm, ne) if aligned else (metric[b * mbs:], mrg[b], unm[b], 1)
            if affine is not None:
                L.tcl_tome_match_affine_f16(mt, mbs, Bt, C, a_pos, na, b_pos, nb, r, affine[0], affine[1], affine[2], mo, uo, self._ws, stream())
            else:
                L.tcl_tome_match_f16(mt, mbs, Bt, C, a_pos, na, b_pos, nb, r, mo, uo, self._ws, stream())
        return mrg, unm, na - r + nb


    # ---- round 6: the chain without `cat` and without a second normalisation
    def _local_len(self, F, N):
        """Tokens a single-round chunk of F frames keeps after the local merge (merge.py:90: r = min(na, int(na * ratio)))."""
        if F <= 1:
            return N
        na = (F - 1) * N
        return na - min(na, int(na * self.args["local_merge_ratio"])) + N

    def _slots(self, coin, TL, Tb):
        """(src_len, loff, boff) of the global match's [src | dst] block (patch.py:61-70)."""
        return (TL, 0, TL) if coin > self.args["global_rand"] else (Tb, Tb, 0)

    def _next_block(self, ne, TL_bank, N, C):
        """If the chunk that will meet the bank this chunk leaves is known (the next one of the pass), allocate ITS [src | dst] token and metric blocks now
        and return where the bank goes in them: (cat, catm, TL_next, loff, boff).  The bank is then written once, into place."""
        lst, i = self._cur
        if lst is None or i + 1 >= len(lst):
            return None
        Fn, _, coin_n = lst[i + 1]
        if Fn > self.args["target_stride"]:
            return None
        TLn = self._local_len(Fn, N)
        _, loff, boff = self._slots(coin_n, TLn, TL_bank)
        T = TLn + TL_bank
        return (torch.empty(ne, T, C, dtype=H16, device=self.dev), torch.empty(ne, T, C, dtype=H16, device=self.dev), TLn, loff, boff)

    def _compute_merge_carried(self, name, x, F, N, C, xbs, metric, ne, lazy_merged):
        """compute_merge for the shape every TC-Light configuration has (one local round, maps shared by the batch entries, global merge on), with the
        cosine-normalised rows CARRIED beside the tokens instead of re-derived: normalisation is per row, so `normalise(gather(x))` == `gather(normalise(x))`
        bit for bit, and the metric of the [local | bank] block is the two gathers the tokens take anyway (tcl_gather_rows_pair_f16: one launch moves
        both).  The block itself is never copied together: the local survivors are gathered straight into their slot of it, and the bank was written into
        its slot by the chunk that left it (`_next_block`: the next chunk's frame count and coin were drawn at begin_step, so its layout is known).
        Per chunk and block this drops two copies of T x C, one normalisation of T x C (k_tome_normalize: 4.6 s of side-stream kernel time per 300-frame
        pass in round 5) and three launches.  Same maps, tokens and banks as the legacy chain, bit for bit (tests/test_gpu_kernels.py, TCL_TOME_CAT=1)."""
        a, L = self.args, self.L
        FN, TL = F * N, self._local_len(F, N)
        if metric is None:                                                      # norm1 did not write it (direct callers / TCL_LN_METRIC=0)
            mx, mbs = torch.empty(ne, FN, C, dtype=H16, device=self.dev), FN * C
            flat = x.reshape(-1)
            for b in range(ne):
                L.tcl_tome_normalize_f16(flat[b * xbs:], mx[b], FN, C, stream())
        else:
            mx, mbs = metric, xbs
        mrg1 = unm1 = None
        if F > 1:
            a_pos, b_pos = self._positions(F, N, self.randf, 0)
            mrg1, unm1, Tn = self._match(None, FN, C, a_pos, a_pos.numel(), b_pos, b_pos.numel(), a["local_merge_ratio"], tbs=mbs,
                                         affine=(self.randf * N, N, self.randf * N), metric=mx, ne=ne)
            assert Tn == TL
        bank, bmet, home = self.banks.get(name), self._bank_met.get(name), self._home.pop(name, None)
        if bank is None:                                                        # patch.py:81-82: the first chunk seeds the bank with its local tokens
            nxt = self._next_block(ne, TL, N, C)
            if nxt is not None:
                cat_n, catm_n, TLn, loff_n, boff_n = nxt
                tok, met = cat_n[:, boff_n:boff_n + TL], catm_n[:, boff_n:boff_n + TL]
                self._home[name] = (cat_n, catm_n, TLn, loff_n, boff_n)
            else:
                tok, met = torch.empty(ne, TL, C, dtype=H16, device=self.dev), torch.empty(ne, TL, C, dtype=H16, device=self.dev)
            L.tcl_gather_rows_pair_f16(x, xbs, mx, mbs, mrg1 if mrg1 is not None else 0, tok, tok.stride(0), met, met.stride(0), ne, TL, C, stream())
            self.banks[name], self._bank_met[name] = tok, met
            if self.trace is not None:
                self.trace.append(dict(name=name, F=F, unm=unm1, gather=mrg1, mrg1=mrg1, T=TL))
            merged = (tok, tok.stride(0), None) if lazy_merged else tok.contiguous()
            return merged, unm1, TL
        if bank.shape[0] != ne:
            raise RuntimeError(f"VidToMe bank of block {name!r} was seeded with {bank.shape[0]} batch entr{'y' if bank.shape[0] == 1 else 'ies'}, this call "
                               f"carries {ne}: call reset_global_tokens() when switching between forward_many(cfg_pair=True) and the two-entry paths")
        Tb = bank.shape[1]
        src_len, loff, boff = self._slots(self.coin, TL, Tb)
        T = TL + Tb
        if home is not None and home[0].shape == (ne, T, C) and home[2:] == (TL, loff, boff) and bank.data_ptr() == home[0][:, boff:].data_ptr():
            cat, catm = home[0], home[1]                                        # the bank (and its metric) already sit in their slot
        else:
            cat, catm = torch.empty(ne, T, C, dtype=H16, device=self.dev), torch.empty(ne, T, C, dtype=H16, device=self.dev)
            if bmet is not None:
                L.tcl_gather_rows_pair_f16(bank, bank.stride(0), bmet, bmet.stride(0), 0, cat[:, boff:], T * C, catm[:, boff:], T * C, ne, Tb, C, stream())
            else:
                L.tcl_gather_rows_pair_f16(bank, bank.stride(0), 0, 0, 0, cat[:, boff:], T * C, 0, 0, ne, Tb, C, stream())
                for b in range(ne):
                    L.tcl_tome_normalize_f16(bank[b], catm[b, boff:], Tb, C, stream())
        L.tcl_gather_rows_pair_f16(x, xbs, mx, mbs, mrg1 if mrg1 is not None else 0, cat[:, loff:], T * C, catm[:, loff:], T * C, ne, TL, C, stream())
        mrg2, unm2, Tm = self._match(None, T, C, self._range(0, src_len), src_len, self._range(src_len, T), T - src_len, a["global_merge_ratio"],
                                     tbs=T * C, affine=(src_len, 0, src_len), metric=catm, ne=ne)
        if lazy_merged:
            merged = (cat, T * C, mrg2)
        else:
            merged = torch.empty(ne, Tm, C, dtype=H16, device=self.dev)
            self._gather(cat, T * C, mrg2, merged, Tm * C, Tm, C, ne)
        unm = self._compose(unm2, unm1, loff, FN)                               # 2s-unmerge then the randframe unmerge (func_warper(u_ls[::-1]))
        bmap = self._compose(mrg2, unm2[loff:], 0, TL)                          # bank <- u(merged_tokens), local part (patch.py:80)
        nxt = self._next_block(ne, TL, N, C)
        if nxt is not None:
            cat_n, catm_n, TLn, loff_n, boff_n = nxt
            tok, met = cat_n[:, boff_n:boff_n + TL], catm_n[:, boff_n:boff_n + TL]
            self._home[name] = (cat_n, catm_n, TLn, loff_n, boff_n)
        else:
            tok, met = torch.empty(ne, TL, C, dtype=H16, device=self.dev), torch.empty(ne, TL, C, dtype=H16, device=self.dev)
        L.tcl_gather_rows_pair_f16(cat, T * C, catm, T * C, bmap, tok, tok.stride(0), met, met.stride(0), ne, TL, C, stream())
        self.banks[name], self._bank_met[name] = tok, met
        if self.trace is not None:
            self.trace.append(dict(name=name, F=F, unm=unm, mrg2=mrg2, mrg1=mrg1, T=Tm, loff=loff, boff=boff, TL=TL, bmap=bmap))
        return merged, unm, Tm

    # ---- patch.py:14-91
    def merges(self, N):
        """patch.py:15-18: does a block with N tokens per frame merge at all?"""
        if not self.enabled:
            return False
        return int(math.ceil(math.sqrt((self.size[0] * self.size[1]) // N))) <= self.args["max_downsample"]

    def compute_merge(self, name, x, F, N, C, _unused=None, xbs=None, metric=None, ne=2, lazy_merged=False):
        """x: norm1 output of one chunk: the unconditional [F*N, C] rows at x, the conditional ones xbs elements further (default
        F*N*C: a contiguous [2F, N, C] == joined [2, F*N, C]); metric: x's cosine-normalised rows in the same layout if norm1 wrote them
        (unet.py: tcl_layernorm_metric_f16), else None.  ne = 1: x holds ONE entry because the caller knows the pair to be identical (the
        classifier-free-guidance pair before the first text cross-attention, unet.py): with equal scores in both entries the matching picks
        the first one (lowest concatenated dst index), i.e. exactly the single-entry matching -- same maps, merged tokens and bank [1, T, C].
        Returns None when this block is not merged, else
        (merged [2,T,C], unm int32 [F*N] ([2, F*N] without align_batch) or None for identity, T).
        lazy_merged=True (unet.py, round 5): where the merged sequence is a gather of the [local | bank] block by a map shared by the entries, it is NOT
        materialised -- `merged` is then the tuple (block [ne, Tcat, C], Tcat*C, map int32 [T]) for a consumer that applies the map in its operand load
        (tcl_gemm_qkv_panels_f16)."""
        a = self.args
        if not self.merges(N):
            return None
        L = self.L
        if xbs is None:
            xbs = F * N * C
        if a["align_batch"] and a["merge_global"] and F <= a["target_stride"] and os.environ.get("TCL_TOME_CAT", "0") == "0":
            return self._compute_merge_carried(name, x, F, N, C, xbs, metric, ne, lazy_merged)
        self._home.pop(name, None)
        self._bank_met.pop(name, None)
        # ---- local merging (patch.py:36-58): randframe rounds until one frame is left -- one round for F <= target_stride (every TC-Light
        # config), 8 -> 2 -> 1 / 16 -> 4 -> 1 for longer chunks, the unmerged tokens of a round riding along as extra dst tokens of the next
        mrg1 = unm1 = None
        seq, sbs, T, unm_pre = x, xbs, F * N, 0
        for cur, randf in zip(*self.round_frames(F, self.randfs)):
            a_pos, b_pos = self._positions(cur, N, randf, unm_pre)
            single = unm_pre == 0 and cur <= a["target_stride"]                # dst = the N tokens of frame randf, src = the rest: affine
            mrg, unm, Tn = self._match(seq, T, C, a_pos, a_pos.numel(), b_pos, b_pos.numel(), a["local_merge_ratio"], tbs=sbs,
                                       affine=(randf * N, N, randf * N) if single else None, metric=metric if seq is x else None, ne=ne)
            nxt = torch.empty(ne, Tn, C, dtype=H16, device=self.dev)
            self._gather(seq, sbs, mrg, nxt, Tn * C, Tn, C, ne)
            if mrg1 is None:
                mrg1, unm1 = mrg, unm
            else:
                mrg1 = self._compose(mrg1, mrg, 0, Tn)                          # merged slot -> row of the joined input
                unm1 = self._compose(unm, unm1, 0, F * N)                       # joined position -> slot of the current sequence
            unm_pre += Tn - b_pos.numel()                                       # ret_dict["unm_num"] = na - r
            seq, sbs, T = nxt, Tn * C, Tn
        if F > 1:
            local, TL = seq, T
        else:
            TL = N                                       # F == 1: nothing to merge locally; keep the [2, T, C] layout
            if xbs == N * C or ne == 1:
                local = x.reshape(-1)[:ne * N * C].view(ne, N, C)
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
        if bank.shape[0] != ne:       # (a real exception: `python -O` strips asserts and a 2-entry gather would then read past a 1-entry bank)
            raise RuntimeError(f"VidToMe bank of block {name!r} was seeded with {bank.shape[0]} batch entr{'y' if bank.shape[0] == 1 else 'ies'}, this call "
                               f"carries {ne}: call reset_global_tokens() when switching between forward_many(cfg_pair=True) and the two-entry paths")
        Tb = bank.shape[1]
        if self.coin > a["global_rand"]:                                        # patch.py:61-65: local tokens are src
            src_len, loff, boff = TL, 0, TL
        else:                                                                   # patch.py:66-70: bank tokens are src
            src_len, loff, boff = Tb, Tb, 0
        T = TL + Tb
        cat = torch.empty(ne, T, C, dtype=H16, device=self.dev)
        L.tcl_gather_rows_f16(local, TL * C, 0, 0, 0, cat[:, loff:], T * C, ne, TL, C, stream())
        L.tcl_gather_rows_f16(bank, bank.stride(0), 0, 0, 0, cat[:, boff:], T * C, ne, Tb, C, stream())
        mrg2, unm2, Tm = self._match(cat, T, C, self._range(0, src_len), src_len, self._range(src_len, T), T - src_len,
                                     a["global_merge_ratio"], affine=(src_len, 0, src_len), ne=ne)
        if lazy_merged and mrg2.dim() == 1:
            merged = (cat, T * C, mrg2)
        else:
            merged = torch.empty(ne, Tm, C, dtype=H16, device=self.dev)
            self._gather(cat, T * C, mrg2, merged, Tm * C, Tm, C, ne)
        unm = self._compose(unm2, unm1, loff, F * N)                            # 2s-unmerge then the randframe unmerges (func_warper(u_ls[::-1]))
        bmap = self._compose(mrg2, unm2[..., loff:].contiguous() if unm2.dim() == 2 else unm2[loff:], 0, TL)     # bank <- u(merged_tokens) (patch.py:80)
        nb_ = torch.empty(ne, TL, C, dtype=H16, device=self.dev)
        self._gather(cat, T * C, bmap, nb_, TL * C, TL, C, ne)
        self.banks[name] = nb_
        if self.trace is not None:
            self.trace.append(dict(name=name, F=F, unm=unm, mrg2=mrg2, mrg1=mrg1, T=Tm, loff=loff, boff=boff, TL=TL, bmap=bmap))
        return merged, unm, Tm
