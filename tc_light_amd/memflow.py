"""MemFlowNet pieces on the device (SURVEY 8(f) rank 2 -- the flow estimator that produces stage-2 inputs).

So far: the correlation block.  `CorrBlock` keeps the reference's interface (utils/evaluation/memflow/core/Networks/MemFlowNet/corr.py:74-120:
`CorrBlock(fmap1, fmap2, num_levels=4, radius=4)(coords) -> [B, L*(2r+1)^2, H, W]`, NCHW f32 in and out) but never builds the all-pairs
volume: windows are computed on demand from an avg-pooled fmap2 pyramid (`tcl_corr_lookup_f32`, csrc/flow.hip).  The encoders, the
GMA / SK update block and the memory read are not ported yet.
"""
import ctypes

import torch

from .lib import lib, stream


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if not fmap1.is_cuda:
            raise RuntimeError("tc_light_amd.memflow.CorrBlock needs device tensors (no CPU path)")
        self.num_levels, self.radius = num_levels, radius
        B, D, H, W = fmap1.shape
        self.shape = (B, D, H, W)
        L = lib()
        self.f1 = fmap1.float().permute(0, 2, 3, 1).contiguous()
        lv = [fmap2.float().permute(0, 2, 3, 1).contiguous()]
        hs, ws = [H], [W]
        for _ in range(num_levels - 1):
            h, w = hs[-1] // 2, ws[-1] // 2
            if h < 1 or w < 1:
                raise ValueError("feature map too small for the requested number of pyramid levels")
            nxt = torch.empty(B, h, w, D, dtype=torch.float32, device=fmap1.device)
            L.tcl_avgpool2_nhwc_f32(lv[-1], nxt, B, hs[-1], ws[-1], D, stream())
            lv.append(nxt); hs.append(h); ws.append(w)
        self.levels = lv
        self._ptrs = (ctypes.c_void_p * num_levels)(*[t.data_ptr() for t in lv])
        self._hs = (ctypes.c_int * num_levels)(*hs)
        self._ws = (ctypes.c_int * num_levels)(*ws)

    def __call__(self, coords):
        B, D, H, W = self.shape
        n = 2 * self.radius + 1
        out = torch.empty(B, self.num_levels * n * n, H, W, dtype=torch.float32, device=coords.device)
        lib().tcl_corr_lookup_f32(self.f1, self._ptrs, self._hs, self._ws, self.num_levels, coords.float().contiguous(), out, B, H, W, D,
                                  self.radius, 1, stream())
        return out
