"""MemFlowNet inference on the device (SURVEY 8(f) rank 2 -- the flow estimator that produces the stage-2 inputs).

`MemFlowEngine` = the reference's MemFlowNet (things_memflownet: BasicEncoder fnet / cnet, GMA-SK2 update block, `InferenceCore` + `MemoryManager`,
utils/evaluation/memflow/...; inference_core_skflow.py:20-54) on f16 NHWC activations: 1x1 convolutions are `tcl_gemm_f16` calls, 3x3 ones the
implicit-GEMM kernel, the rest lives in csrc/flow.hip (7x7 stem, InstanceNorm, depthwise 15x15 / 7x7 convolutions fused with residual + GELU, convex
up-sampling), the memory read runs on the flash kernel (head_dim 128).  `CorrBlock` keeps the reference's interface
(core/Networks/MemFlowNet/corr.py:74-120: `CorrBlock(fmap1, fmap2, num_levels=4, radius=4)(coords) -> [B, L*(2r+1)^2, H, W]`, NCHW f32 in and out)
but never builds the all-pairs volume: windows are computed on demand from an avg-pooled fmap2 pyramid (`tcl_corr_lookup_f32`).
`estimate_flows` reproduces VideoDataParser.load_flow (interleaved future / past pairs on one inference core, per-direction warm start).
Pinned against the reference modules by tests/golden/memflow_*.npz (tests/test_gpu_memflow.py).
"""
import ctypes
import os

import numpy as np
import torch

from .lib import lib, stream

H16 = torch.float16


# ------------------------------------------------------------------------------------------------ parameter shapes / seeded stand-ins
def encoder_param_shapes(prefix, norm, output_dim=256):
    """State-dict keys of BasicEncoder(output_dim, norm_fn) (cnn.py:124-189) under `prefix` ('fnet.' / 'cnet.')."""
    sh = {}

    def nrm(p, c):
        if norm == "batch":
            for k in ("weight", "bias", "running_mean", "running_var"):
                sh[p + k] = (c,)
            sh[p + "num_batches_tracked"] = ()

    sh[prefix + "conv1.weight"] = (64, 3, 7, 7); sh[prefix + "conv1.bias"] = (64,)
    nrm(prefix + "norm1.", 64)
    cin = 64
    for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), 1):
        for bi, (ci, st) in enumerate(((cin, stride), (dim, 1))):
            p = f"{prefix}layer{li}.{bi}."
            sh[p + "conv1.weight"] = (dim, ci, 3, 3); sh[p + "conv1.bias"] = (dim,)
            sh[p + "conv2.weight"] = (dim, dim, 3, 3); sh[p + "conv2.bias"] = (dim,)
            nrm(p + "norm1.", dim); nrm(p + "norm2.", dim)
            if st != 1:
                nrm(p + "norm3.", dim)
                sh[p + "downsample.0.weight"] = (dim, ci, 1, 1); sh[p + "downsample.0.bias"] = (dim,)
                nrm(p + "downsample.1.", dim)                      # the same module as norm3 (cnn.py:42-43): duplicated keys
        cin = dim
    sh[prefix + "conv2.weight"] = (output_dim, 128, 1, 1); sh[prefix + "conv2.bias"] = (output_dim,)
    return sh


def pcblock_param_shapes(p, cin, cout, k_conv):
    """PCBlock4_Deep_nopool_res(C_in, C_out, k_conv) (sk2.py:6-22)."""
    mid = int(1.5 * cin)
    sh = {}
    for i, k in enumerate(k_conv):
        sh[p + f"conv_list.{i}.weight"] = (cin, 1, k, k); sh[p + f"conv_list.{i}.bias"] = (cin,)
    sh[p + "ffn1.0.weight"] = (mid, cin, 1, 1); sh[p + "ffn1.0.bias"] = (mid,)
    sh[p + "ffn1.2.weight"] = (cin, mid, 1, 1); sh[p + "ffn1.2.bias"] = (cin,)
    sh[p + "pw.weight"] = (cin, cin, 1, 1); sh[p + "pw.bias"] = (cin,)
    sh[p + "ffn2.0.weight"] = (mid, cin, 1, 1); sh[p + "ffn2.0.bias"] = (mid,)
    sh[p + "ffn2.2.weight"] = (cout, mid, 1, 1); sh[p + "ffn2.2.bias"] = (cout,)
    return sh


K_CONV, K_GRU = (1, 15), (1, 7)                                  # sk2.py:201-202


def memflow_param_shapes():
    """State-dict keys of MemFlowNet(cfg = things_memflownet: basicencoder cnet/fnet, GMA-SK2) (MemFlow.py:21-64)."""
    sh = {}
    sh.update(encoder_param_shapes("cnet.", "batch"))
    sh.update(encoder_param_shapes("fnet.", "instance"))
    u = "update_block."
    sh.update(pcblock_param_shapes(u + "encoder.convc1.", 324, 256, K_CONV))
    sh.update(pcblock_param_shapes(u + "encoder.convc2.", 256, 192, K_CONV))
    sh[u + "encoder.convf1.weight"] = (128, 2, 1, 1); sh[u + "encoder.convf1.bias"] = (128,)
    sh.update(pcblock_param_shapes(u + "encoder.convf2.", 128, 64, K_CONV))
    sh.update(pcblock_param_shapes(u + "encoder.conv.", 256, 126, K_CONV))
    sh.update(pcblock_param_shapes(u + "gru.", 512, 128, K_GRU))
    sh.update(pcblock_param_shapes(u + "flow_head.", 128, 2, K_CONV))
    sh[u + "mask.0.weight"] = (256, 128, 3, 3); sh[u + "mask.0.bias"] = (256,)
    sh[u + "mask.2.weight"] = (576, 256, 1, 1); sh[u + "mask.2.bias"] = (576,)
    sh[u + "aggregator.to_v.weight"] = (128, 128, 1, 1); sh[u + "aggregator.gamma"] = (1,)
    sh["att.to_qk.weight"] = (256, 128, 1, 1)
    sh["att.pos_emb.rel_height.weight"] = (319, 128); sh["att.pos_emb.rel_width.weight"] = (319, 128)
    sh["att.pos_emb.rel_ind"] = (160, 160)
    return sh


def seeded_state_dict(shapes, seed):
    """Seeded stand-in weights (no checkpoint in the image): He-scaled convs, BatchNorm statistics near identity; the duplicated
    `downsample.1.*` entries repeat `norm3.*`."""
    g = np.random.default_rng(seed)
    sd = {}
    for k, s in shapes.items():
        if ".downsample.1." in k:
            continue
        if k.endswith("rel_ind"):                                               # RelPosEmb buffer (gma.py:16-18), unused at inference
            n = s[0]
            sd[k] = torch.arange(n).view(1, -1) - torch.arange(n).view(-1, 1) + n - 1
        elif k.endswith("gamma"):                                               # zero-initialised in the reference; non-zero so the memory read matters
            sd[k] = torch.tensor([0.5])
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(1, dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy((0.5 + g.random(s)).astype(np.float32))
        elif k.endswith("running_mean") or k.endswith("bias"):
            sd[k] = torch.from_numpy(((0.002 if "flow_head.ffn2.2" in k else 0.1) * g.standard_normal(s)).astype(np.float32))
        elif len(s) == 1:                                                      # norm weight
            sd[k] = torch.from_numpy((0.8 + 0.4 * g.random(s)).astype(np.float32))
        else:
            fan = int(np.prod(s[1:]))
            gain = (1.0 / fan) ** 0.5 if (".ffn" in k or ".pw." in k or ".conv_list." in k) else (2.0 / fan) ** 0.5   # residual branches: keep 15 iterations tame
            if "flow_head.ffn2.2" in k:
                gain *= 0.02                                                    # small flow updates: the 15 GRU iterations stay bounded
            sd[k] = torch.from_numpy((g.standard_normal(s) * gain).astype(np.float32))
    for k in shapes:
        if ".downsample.1." in k:
            sd[k] = sd[k.replace(".downsample.1.", ".norm3.")]
    return sd


def _pad_to(t, dim, n):
    if t.shape[dim] == n:
        return t
    shp = list(t.shape); shp[dim] = n - t.shape[dim]
    return torch.cat([t, torch.zeros(shp, dtype=t.dtype)], dim)


def _up64(c):
    return (c + 63) // 64 * 64


class EncoderEngine:
    """BasicEncoder (cnn.py:124-216) in f16 NHWC: 7x7 stem kernel, implicit-GEMM 3x3 convs, 1x1 convs as GEMMs.  norm 'batch' (cnet):
    eval BatchNorm folded into the conv weights / bias, ReLU in the GEMM epilogue; norm 'instance' (fnet): tcl_instnorm_f16.  Channel
    counts are padded to multiples of 64 with zero weights (96 -> 128); padded channels stay exactly 0 through norm and ReLU."""

    def __init__(self, sd, prefix, norm, device):
        self.dev, self.norm, self.L = torch.device(device), norm, lib()
        f = {k[len(prefix):]: v.float() for k, v in sd.items() if k.startswith(prefix) and v.dtype.is_floating_point}

        def fold(wk, bk, nk):
            w, b = f[wk], f[bk]
            if norm == "batch":
                s = f[nk + "weight"] / torch.sqrt(f[nk + "running_var"] + 1e-5)
                w = w * s.view(-1, 1, 1, 1)
                b = (b - f[nk + "running_mean"]) * s + f[nk + "bias"]
            return w, b

        w, b = fold("conv1.weight", "conv1.bias", "norm1.")
        self.stem = (w.reshape(64, 147).t().contiguous().to(self.dev), b.contiguous().to(self.dev))
        self.blocks = []
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            for bi in (0, 1):
                p = f"layer{li}.{bi}."
                st = stride if bi == 0 else 1
                blk = dict(stride=st)
                for cn, nn_ in (("conv1", "norm1."), ("conv2", "norm2.")):
                    w, b = fold(p + cn + ".weight", p + cn + ".bias", p + nn_)
                    co, ci = _up64(w.shape[0]), _up64(w.shape[1])
                    w = _pad_to(_pad_to(w, 0, co), 1, ci)
                    blk[cn] = (w.permute(0, 2, 3, 1).reshape(co, 9 * ci).to(H16).contiguous().to(self.dev), _pad_to(b, 0, co).to(H16).to(self.dev), ci, co)
                if st != 1:
                    w, b = fold(p + "downsample.0.weight", p + "downsample.0.bias", p + "norm3.")
                    co, ci = _up64(w.shape[0]), _up64(w.shape[1])
                    w = _pad_to(_pad_to(w, 0, co), 1, ci)
                    blk["down"] = (w.reshape(co, ci).to(H16).contiguous().to(self.dev), _pad_to(b, 0, co).to(H16).to(self.dev))
                self.blocks.append(blk)
        self.out = (f["conv2.weight"].reshape(-1, 128).to(H16).contiguous().to(self.dev), f["conv2.bias"].to(H16).to(self.dev))
        self._ws = {}

    def _inorm(self, x, B, HW, C, relu):
        key = (B, C)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = torch.empty(self.L.tcl_instnorm_workspace_bytes(B, C), dtype=torch.uint8, device=self.dev)
        y = torch.empty_like(x)
        self.L.tcl_instnorm_f16(x, y, B, HW, C, 1e-5, int(relu), ws, stream())
        return y

    def _conv3(self, x, B, H, W, spec, stride, relu):
        w, b, ci, co = spec
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(B * Ho * Wo, co, dtype=H16, device=self.dev)
        self.L.tcl_conv3x3_f16(x, w, b, 0, y, B, H, W, ci, co, stride, 1, 0, 0, 3 if relu else 0, stream())
        return y, Ho, Wo

    @torch.no_grad()
    def forward(self, img):
        """img [B,3,H,W] f32 (H, W multiples of 8) -> feature map [B*(H/8)*(W/8), 256] f16 (NHWC rows), (H/8, W/8)."""
        L, bn = self.L, self.norm == "batch"
        B, _, H, W = img.shape
        h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        x = torch.empty(B * h * w, 64, dtype=H16, device=self.dev)
        L.tcl_conv7x7s2_c3_f16(img.float().contiguous(), self.stem[0], self.stem[1], x, B, H, W, int(bn), stream())
        if not bn:
            x = self._inorm(x, B, h * w, 64, True)
        for blk in self.blocks:
            st = blk["stride"]
            y, h2, w2 = self._conv3(x, B, h, w, blk["conv1"], st, bn)
            C = blk["conv1"][3]
            if not bn:
                y = self._inorm(y, B, h2 * w2, C, True)
            y, _, _ = self._conv3(y, B, h2, w2, blk["conv2"], 1, bn)
            if not bn:
                y = self._inorm(y, B, h2 * w2, C, True)
            if st != 1:
                Cin = x.shape[1]
                xs = torch.empty(B * h2 * w2, Cin, dtype=H16, device=self.dev)
                L.tcl_subsample2_nhwc_f16(x, xs, B, h, w, Cin, stream())
                wd, bd = blk["down"]
                xd = torch.empty(B * h2 * w2, C, dtype=H16, device=self.dev)
                L.tcl_gemm_f16(xs, wd, bd, 0, xd, B * h2 * w2, C, Cin, Cin, Cin, C, C, 0, stream())
                x = xd if bn else self._inorm(xd, B, h2 * w2, C, False)
            out = torch.empty_like(y)
            L.tcl_add_act_f16(x, y, out, y.numel(), 3, stream())
            x, h, w = out, h2, w2
        wo, bo = self.out
        f = torch.empty(B * h * w, wo.shape[0], dtype=H16, device=self.dev)
        L.tcl_gemm_f16(x, wo, bo, 0, f, B * h * w, wo.shape[0], 128, 128, 128, wo.shape[0], wo.shape[0], 0, stream())
        return f, (h, w)



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
        # round 6: f16 copies for the tile-sharing lookup (tcl_corr_lookup_rows_tiled_f16): level 0 came out of the encoder in f16, so this is its exact value;
        # the pooled levels are rounded once (f32 pooling chain above, as before)
        self.f1h, self.levels_h = self.f1.half(), [t.half() for t in lv]
        self._ptrs_h = (ctypes.c_void_p * num_levels)(*[t.data_ptr() for t in self.levels_h])
        self._flags = torch.zeros(num_levels * ((H + 7) // 8) * ((W + 7) // 8), dtype=torch.int32, device=fmap1.device)
        self._ptrs = (ctypes.c_void_p * num_levels)(*[t.data_ptr() for t in lv])
        self._hs = (ctypes.c_int * num_levels)(*hs)
        self._ws = (ctypes.c_int * num_levels)(*ws)

    @classmethod
    def from_nhwc(cls, f1, f2, num_levels=4, radius=4):
        """f1, f2 [B,H,W,D] f32 NHWC on the device."""
        return cls(f1.permute(0, 3, 1, 2), f2.permute(0, 3, 1, 2), num_levels, radius)

    def lookup_rows(self, coords, out_rows):
        """Window lookup written as f16 rows [B*H*W, ld] (first L*(2r+1)^2 channels) for the motion encoder's GEMM."""
        B, D, H, W = self.shape
        if B == 1 and self.radius == 4 and os.environ.get("TCL_CORR_TILED", "1") != "0":
            lib().tcl_corr_lookup_rows_tiled_f16(self.f1h, self._ptrs_h, self.f1, self._ptrs, self._hs, self._ws, self.num_levels, coords, out_rows,
                                                 out_rows.shape[1], H, W, D, self.radius, self._flags, stream())
            return
        lib().tcl_corr_lookup_rows_f16(self.f1, self._ptrs, self._hs, self._ws, self.num_levels, coords, out_rows, out_rows.shape[1], B, H, W, D,
                                       self.radius, stream())

    def __call__(self, coords):
        B, D, H, W = self.shape
        n = 2 * self.radius + 1
        out = torch.empty(B, self.num_levels * n * n, H, W, dtype=torch.float32, device=coords.device)
        lib().tcl_corr_lookup_f32(self.f1, self._ptrs, self._hs, self._ws, self.num_levels, coords.float().contiguous(), out, B, H, W, D,
                                  self.radius, 1, stream())
        return out


# ------------------------------------------------------------------------------------------------ update block + inference core
class _Lin:
    """1x1 convolution as a GEMM over channel-padded f16 rows: weight [co64, ci64] (zero rows / columns for the padding), bias [co64]."""

    def __init__(self, w, b, dev):
        co, ci = w.shape[0], w.shape[1]
        self.ci, self.co = _up64(ci), _up64(co)
        self.w = _pad_to(_pad_to(w.reshape(co, ci), 0, self.co), 1, self.ci).to(H16).contiguous().to(dev)
        self.b = _pad_to(b if b is not None else torch.zeros(co), 0, self.co).to(H16).contiguous().to(dev)


class _PCBlock:
    """PCBlock4_Deep_nopool_res (sk2.py:6-30): x = gelu(x + ffn1(x)); x = gelu(x + dw_k(x)) for k in k_conv; x = gelu(x + pw(x)); ffn2(x)."""

    def __init__(self, f, p, k_conv, dev):
        lin = lambda q: _Lin(f[q + "weight"], f[q + "bias"], dev)
        self.f1a, self.f1b, self.pw, self.f2a, self.f2b = lin(p + "ffn1.0."), lin(p + "ffn1.2."), lin(p + "pw."), lin(p + "ffn2.0."), lin(p + "ffn2.2.")
        self.dw = []
        for i, k in enumerate(k_conv):
            w, b = f[p + f"conv_list.{i}.weight"], f[p + f"conv_list.{i}.bias"]
            c = _up64(w.shape[0])
            self.dw.append((_pad_to(w.reshape(w.shape[0], k * k).t(), 1, c).to(H16).contiguous().to(dev), _pad_to(b, 0, c).to(H16).contiguous().to(dev), k))


class MemFlowEngine:
    """MemFlowNet (things_memflownet: basicencoder cnet/fnet, GMA-SK2 update block) + InferenceCore + MemoryManager
    (MemFlow.py:21-183, sk2.py, inference/inference_core_skflow.py:20-54, memory_manager_skflow.py) on the device: f16 NHWC activations
    with channel counts padded to multiples of 64, f32 coordinates / flow / correlation inputs."""

    def __init__(self, state_dict, device, iters=15, train_avg_length=(400 * 720 // 64) * 3 / 2, max_mt=2, min_mt=1):
        missing = [k for k in memflow_param_shapes() if k not in state_dict]
        if missing:
            raise KeyError(f"MemFlowNet state dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        self.dev, self.L = torch.device(device), lib()
        self.iters, self.tal, self.max_mt, self.min_mt = iters, train_avg_length, max_mt, min_mt
        f = {k: v.float() for k, v in state_dict.items() if v.dtype.is_floating_point}
        d = self.dev
        self.fnet = EncoderEngine(state_dict, "fnet.", "instance", d)
        self.cnet = EncoderEngine(state_dict, "cnet.", "batch", d)
        u = "update_block."
        self.convc1, self.convc2 = _PCBlock(f, u + "encoder.convc1.", K_CONV, d), _PCBlock(f, u + "encoder.convc2.", K_CONV, d)
        self.convf1 = _Lin(f[u + "encoder.convf1.weight"], f[u + "encoder.convf1.bias"], d)
        self.convf2, self.conv = _PCBlock(f, u + "encoder.convf2.", K_CONV, d), _PCBlock(f, u + "encoder.conv.", K_CONV, d)
        self.gru, self.flow_head = _PCBlock(f, u + "gru.", K_GRU, d), _PCBlock(f, u + "flow_head.", K_CONV, d)
        w = f[u + "mask.0.weight"]
        self.mask0 = (w.permute(0, 2, 3, 1).reshape(256, 9 * 128).to(H16).contiguous().to(d), f[u + "mask.0.bias"].to(H16).to(d))
        self.mask2 = _Lin(f[u + "mask.2.weight"], f[u + "mask.2.bias"], d)
        self.to_v = _Lin(f[u + "aggregator.to_v.weight"], None, d)
        self.gamma = float(f[u + "aggregator.gamma"].item())
        self.to_qk = _Lin(f["att.to_qk.weight"], None, d)
        self.scale = 128 ** -0.5                                   # Attention.scale, gma.py:47
        self.clear_memory()

    def clear_memory(self):
        self.mem_k = self.mem_v = None                             # rows [T, 128] f16 (working memory, oldest first)

    # ---- building blocks
    def _gemm(self, x, lin, M, act=0, resid=None, lda=None):
        y = torch.empty(M, lin.co, dtype=H16, device=self.dev)
        self.L.tcl_gemm_f16(x, lin.w, lin.b, resid if resid is not None else 0, y, M, lin.co, lin.ci, lda or lin.ci, lin.ci, lin.co, lin.co, act, stream())
        return y

    def _pc(self, blk, x, B, h, w, out_act=0):
        M = B * h * w
        t = self._gemm(x, blk.f1a, M, act=4)
        x = self._gemm(t, blk.f1b, M, act=5, resid=x)
        for wd, bd, k in blk.dw:
            y = torch.empty_like(x)
            self.L.tcl_dwconv_gelu_f16(x, wd, bd, y, B, h, w, x.shape[1], k, stream())
            x = y
        x = self._gemm(x, blk.pw, M, act=5, resid=x)
        t = self._gemm(x, blk.f2a, M, act=4)
        return self._gemm(t, blk.f2b, M, act=out_act)

    def _cat(self, a, b, M):
        y = torch.empty(M, a.shape[1] + b.shape[1], dtype=H16, device=self.dev)
        self.L.tcl_concat_channels_f16(a, a.shape[1], b, b.shape[1], y, M, stream())
        return y

    @torch.no_grad()
    def step(self, images, end=False, flow_init=None):
        """InferenceCore.step: images [1,2,3,H,W] f32 in [-1,1] (H, W multiples of 8; H/8/8 >= 2) -> (flow_low [1,2,H/8,W/8], flow_up [1,2,H,W]).

        Round 6, measured and OFF by default: a frame pair is ~1 050 launches of kernels that take 5-40 us each (P = 14 400 rows at 1280x720), which looked
        launch-bound -- it is not: the kernel table (profiles/r6_memflow_kernel_stats_before.txt) sums to 48.4 ms per pair against 50.4 ms of wall clock; the
        host keeps ahead of the GPU.  With TCL_MEMFLOW_GRAPH=1 the device work of a step is captured ONCE per (image shape, memory length, warm start or not)
        into a HIP graph and replayed (first step of a shape eager, second captured, later ones copy-in -> one graph launch -> copy-out; same kernels, order
        and arguments: same bits, tests/test_gpu_memflow.py::test_step_graph_replay_equals_eager) -- 48.2-48.9 ms per pair against 47.7 eager on one box
        (profiles/r6_ab_memflow_graph_nw.txt): the copies cost what the launch gaps saved.  What a pair spends is kernel time: correlation lookup 26 %, memory-read
        attention 20 %, 15x15 depthwise convolutions 18 %, small GEMMs 21 %."""
        d = self.dev
        images = images.to(d).float().contiguous()
        fi = None if flow_init is None else flow_init.to(d).float().contiguous()
        Tm = 0 if self.mem_k is None else self.mem_k.shape[0]
        use = os.environ.get("TCL_MEMFLOW_GRAPH", "0") == "1" and d.type == "cuda"
        if use:
            key = (tuple(images.shape), Tm, fi is not None)
            graphs = self.__dict__.setdefault("_graphs", {})
            ent = graphs.get(key)
            if ent is None:
                graphs[key] = "seen"                                 # eager this time
                ent = None
            elif ent == "seen":
                ent = graphs[key] = self._capture(images, fi, Tm)
            if isinstance(ent, dict):
                ent["images"].copy_(images)
                if fi is not None:
                    ent["fi"].copy_(fi)
                if Tm:
                    ent["mk"].copy_(self.mem_k); ent["mv"].copy_(self.mem_v)
                ent["graph"].replay()
                flow_low, up, k_all, v_all = (t.clone() for t in ent["out"])
                self._remember(k_all, v_all, end, Tm)
                return flow_low, up
        flow_low, up, k_all, v_all = self._core(images, self.mem_k, self.mem_v, fi)
        self._remember(k_all, v_all, end, Tm)
        return flow_low, up

    def _remember(self, k_all, v_all, end, Tm):
        if not end:                                                # mem_every = 1 (inference_core_skflow.py:25, :49-51)
            self.mem_k, self.mem_v = k_all, v_all
            P = k_all.shape[0] - Tm
            if self.mem_k.shape[0] >= self.max_mt * P:             # compress_features: keep the last min_mt frames
                self.mem_k, self.mem_v = self.mem_k[-self.min_mt * P:].contiguous(), self.mem_v[-self.min_mt * P:].contiguous()

    def _capture(self, images, fi, Tm):
        ent = dict(images=images.clone(), fi=None if fi is None else fi.clone(), mk=self.mem_k.clone() if Tm else None, mv=self.mem_v.clone() if Tm else None)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ent["out"] = self._core(ent["images"], ent["mk"], ent["mv"], ent["fi"])
        ent["graph"] = g
        return ent

    def _core(self, images, mem_k, mem_v, flow_init):
        """The device work of one step: (images, working memory, warm start) -> (flow_low, flow_up, k_all, v_all)."""
        L, d = self.L, self.dev
        # context: net = tanh(c[:, :128]), inp = relu(c[:, 128:]), (query, key) = to_qk(inp)   (MemFlow.py:112-118)
        c, (h, w) = self.cnet.forward(images[:, 0])
        P = h * w
        net = torch.empty(P, 128, dtype=H16, device=d); inp = torch.empty(P, 128, dtype=H16, device=d)
        L.tcl_context_split_f16(c, net, inp, P, stream())
        qk = self._gemm(inp, self.to_qk, P)                        # [P, 256]: query | key
        key = qk[:, 128:].contiguous()
        fm, _ = self.fnet.forward(images[0])
        fm = fm.float().view(2, h, w, 256)
        corr_fn = CorrBlock.from_nhwc(fm[0:1].contiguous(), fm[1:2].contiguous())
        ys, xs = torch.meshgrid(torch.arange(h, device=d).float(), torch.arange(w, device=d).float(), indexing="ij")
        coords0 = torch.stack([xs, ys])[None].contiguous()
        coords1 = coords0.clone() if flow_init is None else (coords0 + flow_init).contiguous()
        corr_rows = torch.zeros(P, 384, dtype=H16, device=d)       # 324 channels + zero padding
        k_all = key if mem_k is None else torch.cat([mem_k, key])
        T = k_all.shape[0]
        scale = self.scale * np.log(T) / np.log(self.tal)          # memory_manager_skflow.py:59 (math.log(T, train_avg_length))
        # round 6: one head and one entry leave the flash kernel 114 blocks at 1280x720 -- the keys are cut into chunks that run as batch entries
        # (tcl_attention_splitkv_f16; TCL_MEMFLOW_SPLITKV = number of chunks, 0 = off)
        ns = int(os.environ.get("TCL_MEMFLOW_SPLITKV", "3"))
        if ns >= 2 and (T % (64 * ns) != 0 or -(-P // 128) * ns > 1024):
            ns = next((c for c in (3, 2, 5, 4, 6) if T % (64 * c) == 0 and -(-P // 128) * c <= 1024), 0)
        wsp = torch.empty(L.tcl_attention_splitkv_workspace_bytes(ns, 1, P, T, 128), dtype=torch.uint8, device=d) if ns >= 2 else None
        if ns < 2:
            wq = torch.empty(L.tcl_attention_q_bytes(1, 1, P, 128), dtype=torch.uint8, device=d)
            wkv = torch.empty(L.tcl_attention_kv_bytes(1, 1, T, 128), dtype=torch.uint8, device=d)
        for it in range(self.iters):
            corr_fn.lookup_rows(coords1, corr_rows)
            flow = coords1 - coords0
            cor = self._pc(self.convc2, self._pc(self.convc1, corr_rows, 1, h, w, out_act=4), 1, h, w)
            frow = torch.empty(P, 64, dtype=H16, device=d)
            L.tcl_nchw_f32_to_rows_f16(flow, frow, 1, 2, P, 64, 0, 1, stream())
            flo = self._pc(self.convf2, self._gemm(frow, self.convf1, P), 1, h, w)
            mf = self._pc(self.conv, self._cat(cor, flo, P), 1, h, w)          # [P,128]: 126 channels + 2 zero
            L.tcl_nchw_f32_to_rows_f16(flow, mf, 1, 2, P, 128, 126, 0, stream())  # torch.cat([out, flow], 1)   (sk2.py:128)
            val = self._gemm(mf, self.to_v, P)
            v_all = val if mem_v is None else torch.cat([mem_v, val])
            ro = torch.empty(P, 128, dtype=H16, device=d)
            if ns >= 2:
                L.tcl_attention_splitkv_f16(qk, 256, k_all, 128, v_all, 128, ro, 128, 1, P, T, 128, float(scale), ns, wsp, stream())
            else:
                L.tcl_attention_f16(qk, 256, P * 256, k_all, 128, T * 128, v_all, 128, T * 128, ro, 128, P * 128, 1, 1, P, T, 128, float(scale), 1, 1, wq, wkv,
                                    stream())
            mfg = torch.empty_like(mf)
            L.tcl_axpy_f16(mf, ro, self.gamma, mfg, mf.numel(), stream())
            net = self._pc(self.gru, self._cat(self._cat(net, inp, P), self._cat(mf, mfg, P), P), 1, h, w)
            delta = self._pc(self.flow_head, net, 1, h, w)          # [P,64]: 2 channels + zero padding
            L.tcl_rows_f16_to_nchw_f32(delta, coords1, 1, 2, P, 64, 0, 1.0, 1.0, stream())
        m1 = torch.empty(P, 256, dtype=H16, device=d)
        L.tcl_conv3x3_f16(net, self.mask0[0], self.mask0[1], 0, m1, 1, h, w, 128, 256, 1, 1, 0, 0, 3, stream())
        mask = self._gemm(m1, self.mask2, P)                       # [P,576]; the 0.25 factor is applied in the upsampling kernel
        flow_low = coords1 - coords0
        up = torch.empty(1, 2, 8 * h, 8 * w, dtype=torch.float32, device=d)
        L.tcl_upsample_flow_f32(flow_low.contiguous(), mask, 576, 0.25, up, 1, h, w, stream())
        return flow_low, up, k_all, v_all


# ------------------------------------------------------------------------------------------------ video-level driver
def forward_interpolate(flow):
    """core/utils/utils.py:32-63 (host, scipy): push every flow vector to where it points and fill by nearest neighbour -- the warm
    start of the next frame pair (video_dataparser.py:154).  flow [2,h,w] tensor -> [2,h,w] f32 CPU tensor."""
    from scipy import interpolate
    f = flow.detach().float().cpu().numpy()
    dx, dy = f[0], f[1]
    ht, wd = dx.shape
    x0, y0 = np.meshgrid(np.arange(wd), np.arange(ht))
    x1, y1 = (x0 + dx).reshape(-1), (y0 + dy).reshape(-1)
    dxr, dyr = dx.reshape(-1), dy.reshape(-1)
    valid = (x1 > 0) & (x1 < wd) & (y1 > 0) & (y1 < ht)
    x1, y1, dxr, dyr = x1[valid], y1[valid], dxr[valid], dyr[valid]
    if len(x1) == 0:
        return torch.zeros(f.shape)
    fx = interpolate.griddata((x1, y1), dxr, (x0, y0), method="nearest", fill_value=0)
    fy = interpolate.griddata((x1, y1), dyr, (x0, y0), method="nearest", fill_value=0)
    return torch.from_numpy(np.stack([fx, fy], axis=0)).float()


def _pad8(x):
    """eval_utils.InputPadder (mode 'sintel'): replicate-pad H, W up to multiples of 8, split evenly."""
    ht, wd = x.shape[-2:]
    ph, pw = (((ht // 8) + 1) * 8 - ht) % 8, (((wd // 8) + 1) * 8 - wd) % 8
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    return torch.nn.functional.pad(x, pad, mode="replicate"), pad


def estimate_flows(engine, frames, warm_start=True):
    """VideoDataParser.load_flow / calc_flow for the 'memflow' model (video_dataparser.py:63-110, 141-156): frames [N,3,H,W] in [0,1] ->
    (future_flows, past_flows) [N,2,H,W] f32 on the device.  As in the reference ONE inference core (one working memory) serves the
    interleaved forward and backward pairs, each direction with its own warm-start chain; the last future flow and the first past flow
    are zero."""
    gts = frames.float() * 2.0 - 1.0
    N = gts.shape[0]
    fut, past = [], []
    prev = {True: None, False: None}
    engine.clear_memory()
    for idx in range(N):
        for is_future in (True, False):
            zero_idx = N - 1 if is_future else 0
            if idx == zero_idx:
                flow = torch.zeros_like(gts[idx:idx + 1, :2])
            else:
                src, tgt = gts[idx:idx + 1], (gts[idx + 1:idx + 2] if is_future else gts[idx - 1:idx])
                (src, pad), (tgt, _) = _pad8(src), _pad8(tgt)
                low, up = engine.step(torch.cat([src, tgt])[None], flow_init=prev[is_future] if warm_start else None)
                ht, wd = up.shape[-2:]
                flow = up[..., pad[2]:ht - pad[3], pad[0]:wd - pad[1]]
                prev[is_future] = forward_interpolate(low[0])[None].to(up.device)
            (fut if is_future else past).append(flow)
    return torch.cat(fut), torch.cat(past)
