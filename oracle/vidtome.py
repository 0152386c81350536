"""Oracle (CPU) for VidToMe token merging.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates utils/VidToMe/vidtome/merge.py:20-159 (bipartite_soft_matching_randframe), :343-463
(bipartite_soft_matching_2s) and patch.py:14-91 (compute_merge) for the configuration TC-Light runs
(align_batch=True, merge_mode "replace", target_stride 4, one local round because F <= 4).
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


def match(metric, a_pos, b_pos, ratio, emulate_f16=False):
    """-> (mrg [na-r+nb], unm [T]) int64 maps; metric [B, T, C]."""
    B, T, _ = metric.shape
    mt = _normalize(metric, emulate_f16)
    a, b = mt[:, a_pos], mt[:, b_pos]
    scores = a @ b.transpose(-1, -2)                       # merge.py:87
    if emulate_f16:
        scores = scores.half().float()
    na, nb = len(a_pos), len(b_pos)
    r = min(na, int(na * ratio))                           # merge.py:90
    cat = torch.cat([*scores], dim=-1)                     # merge.py:96 (align_batch)
    node_max = cat.max(dim=-1).values
    # first index attaining the max (CPU torch.max semantics; made explicit)
    node_idx = (cat == node_max[:, None]).float().argmax(dim=-1)
    order = torch.sort(node_max, descending=True, stable=True).indices
    unm_idx, src_idx = order[r:], order[:r]
    dst_idx = node_idx[src_idx] % nb
    nun = na - r
    mrg = torch.cat([a_pos[unm_idx], b_pos])
    unm = torch.full((T,), -1, dtype=torch.int64)
    unm[b_pos] = nun + torch.arange(nb)
    unm[a_pos[unm_idx]] = torch.arange(nun)
    unm[a_pos[src_idx]] = nun + dst_idx
    return mrg, unm


def randframe_positions(F, N, randf):
    """merge.py:52-66 with unm_pre = 0: dst = every token of frame randf, src = the rest (in order)."""
    idx = torch.arange(F * N)
    dst = (idx // N) % min(4, F) == randf
    return idx[~dst], idx[dst]


def compute_merge(x, F, bank, randf, coin, local_ratio=0.6, global_ratio=0.5, global_rand=0.5,
                  merge_global=True, emulate_f16=False):
    """patch.py:14-91 for one patched block.  x: [2F, N, C] (norm1 output), bank: [2, Tb, C] or None.
    coin: the torch.rand(1) draw of patch.py:61.  Returns dict(merged [2,T,C], unmerge(y)->[2F,N,C], bank_new,
    gather (source code per merged slot: >=0 row of joined x, <0 ~row of bank), unm (per joined position))."""
    B2, N, C = x.shape
    xj = x.reshape(2, F * N, C)                           # join_frame (vidtome/utils.py:32-35)
    if F > 1:
        a_pos, b_pos = randframe_positions(F, N, randf)
        mrg1, unm1 = match(xj, a_pos, b_pos, local_ratio, emulate_f16)
        local = xj[:, mrg1]
    else:
        mrg1 = torch.arange(N)
        unm1 = torch.arange(N)
        local = xj
    TL = local.shape[1]
    if not merge_global or bank is None:
        return dict(merged=local, unm=unm1, gather=mrg1, bank_new=local.clone() if merge_global else None,
                    unmerge=lambda y: y[:, unm1].reshape(B2, N, -1))
    Tb = bank.shape[1]
    if coin > global_rand:                                 # patch.py:61-65 local tokens are src
        tokens = torch.cat([local, bank], 1)
        src_len, loff, boff = TL, 0, TL
    else:                                                  # patch.py:66-70 bank tokens are src
        tokens = torch.cat([bank, local], 1)
        src_len, loff, boff = Tb, Tb, 0
    T = tokens.shape[1]
    mrg2, unm2 = match(tokens, torch.arange(src_len), torch.arange(src_len, T), global_ratio, emulate_f16)
    merged = tokens[:, mrg2]
    unm = unm2[loff + unm1]                                # func_warper(u_ls[::-1]): 2s unmerge (local chunk), then randframe
    cat_pos = mrg2
    gather = torch.where((cat_pos >= loff) & (cat_pos < loff + TL), mrg1[(cat_pos - loff).clamp(0, TL - 1)], -(cat_pos - boff) - 1)
    bank_new = merged[:, unm2[loff:loff + TL]]             # patch.py:80 u(merged_tokens): the local chunk of the unmerge
    return dict(merged=merged, unm=unm, gather=gather, bank_new=bank_new,
                unmerge=lambda y: y[:, unm].reshape(B2, N, -1))
