"""Oracle (CPU) for VidToMe token merging.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates utils/VidToMe/vidtome/merge.py:20-159 (bipartite_soft_matching_randframe), :343-463
(bipartite_soft_matching_2s) and patch.py:14-91 (compute_merge), merge_mode "replace": the configuration TC-Light runs (align_batch=True,
target_stride 4, one local round because F <= 4) and the branches it leaves idle -- the multi-round local merge of longer chunks
(patch.py:43-56: 8 -> 2 -> 1 frames, the unmerged tokens of earlier rounds joining the dst set) and per-sample matching
(align_batch=False, merge.py:109-118).
Random choices (randf, the global coin) are explicit inputs instead of torch.Generator draws.

`emulate_f16=True` reproduces the f16 pipeline of the reference on GPU (metric/norm and scores rounded to
f16) with the deterministic tie rule the HIP kernels use: highest score, then lowest concatenated dst index;
equal node_max keep ascending src order (stable sort).  On CPU f32 inputs without ties this equals the
reference exactly (pinned by tests/golden/vidtome.npz).
"""
import torch


def _normalize(metric, emulate_f16):
    if emulate_f16:
        m = metric.half()
        n = m.float().pow(2).sum(-1, keepdim=True).sqrt().half()
        return (m.float() / n.float()).half().float()
    return metric / metric.norm(dim=-1, keepdim=True)


def _match_one(scores, a_pos, b_pos, r, T):
    """scores [na, nb * k] (k batch entries concatenated along dst when aligned) -> (mrg, unm) for one matching."""
    na, nb = len(a_pos), len(b_pos)
    node_max = scores.max(dim=-1).values
    # first index attaining the max (CPU torch.max semantics; made explicit)
    node_idx = (scores == node_max[:, None]).float().argmax(dim=-1)
    order = torch.sort(node_max, descending=True, stable=True).indices
    unm_idx, src_idx = order[r:], order[:r]
    dst_idx = node_idx[src_idx] % nb                       # merge.py:104-105 (aligned) / :118 (per sample: no-op)
    nun = na - r
    mrg = torch.cat([a_pos[unm_idx], b_pos])
    unm = torch.full((T,), -1, dtype=torch.int64)
    unm[b_pos] = nun + torch.arange(nb)
    unm[a_pos[unm_idx]] = torch.arange(nun)
    unm[a_pos[src_idx]] = nun + dst_idx
    return mrg, unm


def match(metric, a_pos, b_pos, ratio, emulate_f16=False, align_batch=True):
    """metric [B, T, C] -> (mrg [na-r+nb], unm [T]) int64 maps shared by the batch (align_batch, merge.py:93-108) or
    (mrg [B, na-r+nb], unm [B, T]) per sample (merge.py:109-118)."""
    B, T, _ = metric.shape
    mt = _normalize(metric, emulate_f16)
    a, b = mt[:, a_pos], mt[:, b_pos]
    scores = a @ b.transpose(-1, -2)                       # merge.py:87
    if emulate_f16:
        scores = scores.half().float()
    na = len(a_pos)
    r = min(na, int(na * ratio))                           # merge.py:90
    if align_batch:
        return _match_one(torch.cat([*scores], dim=-1), a_pos, b_pos, r, T)       # merge.py:96
    per = [_match_one(scores[i], a_pos, b_pos, r, T) for i in range(B)]
    return torch.stack([m for m, _ in per]), torch.stack([u for _, u in per])


def _take(x, idx):
    """x [B, T, ...] gathered along dim 1 by a shared [L] or per-sample [B, L] index."""
    if idx.dim() == 1:
        return x[:, idx]
    return torch.stack([x[i, idx[i]] for i in range(x.shape[0])])


def _compose(outer, inner):
    """outer[inner] for shared / per-sample maps (func_warper composition)."""
    if outer.dim() == 1 and inner.dim() == 1:
        return outer[inner]
    B = outer.shape[0] if outer.dim() == 2 else inner.shape[0]
    o = outer if outer.dim() == 2 else outer[None].expand(B, -1)
    i = inner if inner.dim() == 2 else inner[None].expand(B, -1)
    return torch.stack([o[k][i[k]] for k in range(B)])


def randframe_positions(F, N, randf, unm_pre=0, target_stride=4):
    """merge.py:44-66: the sequence is [unm_pre | F frames of N tokens]; dst = every token of the frames f with f % min(target_stride, F)
    == randf, followed by the unm_pre leading tokens (merge.py:65-67); src = the other frames' tokens (in order)."""
    idx = torch.arange(F * N)
    dst = (idx // N) % min(target_stride, F) == randf
    return idx[~dst] + unm_pre, torch.cat([idx[dst] + unm_pre, torch.arange(unm_pre)])


def compute_merge(x, F, bank, randf, coin, local_ratio=0.6, global_ratio=0.5, global_rand=0.5,
                  merge_global=True, emulate_f16=False, align_batch=True, target_stride=4):
    """patch.py:14-91 for one patched block.  x: [2F, N, C] (norm1 output), bank: [2, Tb, C] or None.
    randf: the torch.randint draw(s) of merge.py:56-58 -- an int, or one per randframe round for F > target_stride (8 -> 2 -> 1);
    coin: the torch.rand(1) draw of patch.py:61.  Returns dict(merged [2,T,C], unmerge(y)->[2F,N,C], bank_new,
    gather (source code per merged slot: >=0 row of joined x, <0 ~row of bank), unm (per joined position), bank_src (merged slot per row
    of bank_new)); maps are 1-D when the batch shares them (align_batch) and [2, .] otherwise."""
    B2, N, C = x.shape
    xj = x.reshape(2, F * N, C)                           # join_frame (vidtome/utils.py:32-35)
    randfs = list(randf) if isinstance(randf, (list, tuple)) else [randf]
    seq, unm_pre, cur, rnd = xj, 0, F, 0
    mrg1 = unm1 = torch.arange(F * N)
    while cur > 1:                                        # patch.py:44-56
        a_pos, b_pos = randframe_positions(cur, N, randfs[rnd], unm_pre, target_stride)
        mrg, unm = match(seq, a_pos, b_pos, local_ratio, emulate_f16, align_batch)
        r = min(len(a_pos), int(len(a_pos) * local_ratio))
        seq = _take(seq, mrg)
        mrg1 = _compose(mrg1, mrg)                        # merged slot -> row of the joined input
        unm1 = _compose(unm, unm1)                        # joined position -> slot of the current sequence
        unm_pre += len(a_pos) - r                         # ret_dict["unm_num"]
        cur = (seq.shape[1] - unm_pre) // N
        rnd += 1
    local = seq
    TL = local.shape[1]

    def unmerge_with(umap):
        return lambda y: _take(y, umap).reshape(B2, N, -1)
    if not merge_global or bank is None:
        return dict(merged=local, unm=unm1, gather=mrg1, bank_new=local.clone() if merge_global else None, unmerge=unmerge_with(unm1), rounds=rnd,
                    bank_src=torch.arange(TL))
    Tb = bank.shape[1]
    if coin > global_rand:                                 # patch.py:61-65 local tokens are src
        tokens = torch.cat([local, bank], 1)
        src_len, loff, boff = TL, 0, TL
    else:                                                  # patch.py:66-70 bank tokens are src
        tokens = torch.cat([bank, local], 1)
        src_len, loff, boff = Tb, Tb, 0
    T = tokens.shape[1]
    mrg2, unm2 = match(tokens, torch.arange(src_len), torch.arange(src_len, T), global_ratio, emulate_f16, align_batch)
    merged = _take(tokens, mrg2)
    unm = _compose(unm2, loff + unm1)                      # func_warper(u_ls[::-1]): 2s unmerge (local chunk), then the randframe rounds
    local_rows = _compose(mrg1, (mrg2 - loff).clamp(0, TL - 1))
    gather = torch.where((mrg2 >= loff) & (mrg2 < loff + TL), local_rows, -(mrg2 - boff) - 1)
    bank_new = _take(merged, unm2[..., loff:loff + TL])    # patch.py:80 u(merged_tokens): the local chunk of the unmerge
    return dict(merged=merged, unm=unm, gather=gather, bank_new=bank_new, unmerge=unmerge_with(unm), rounds=rnd,
                bank_src=unm2[..., loff:loff + TL])     # merged slot each row of bank_new was taken from
