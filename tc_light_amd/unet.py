"""IC-Light / SD-1.5 UNet forward on the HIP C ABI, with VidToMe token merging in every transformer block.

Host-side mirror of what the reference assembles from diffusers + monkey-patching:
  UNet2DConditionModel.forward (diffusers 0.32.1; call site generate.py:342-347), the 8-channel conv_in and the
  concat_conds hook (utils/model_utils.py:21-40), ToMeBlock.forward (utils/VidToMe/vidtome/patch.py:124-201).
Activations are NHWC f16 [B, H*W, C]; every op below is one C-ABI call (tc_light_amd/csrc/*.hip).  Python only
sequences launches and owns buffers; there is no torch arithmetic on the data path.
"""
import math
import os

import torch

from .lib import lib, stream
from . import sd15
from .vidtome import VidToMe

H16 = torch.float16


def _dev(t, dev):
    return t.to(device=dev, dtype=H16).contiguous()


def _geglu_rows(w):
    """ff.net.0.proj rows [value (D) | gate (D)] -> 64-row groups [32 value | 32 matching gate] for the fused GEGLU epilogue."""
    D = w.shape[0] // 2
    idx = torch.arange(D).reshape(-1, 32)
    return w[torch.cat([idx, idx + D], dim=1).reshape(-1)]


def _conv_w(w, dev):          # [Cout,Cin,3,3] -> [Cout, 9*Cin] tap-major
    return _dev(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), dev)


GEMM_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tune_gfx950.txt")


def load_gemm_table(L):
    """The committed GEMM tile table (shape -> tile configuration, measured on an MI355X by `bench.py --save_gemm_table`): loaded once per
    process, so known shapes never pay the in-call autotuner (no host sync in the hot loop, same tile run to run).  TCL_GEMM_TABLE overrides
    the path; TCL_AUTOTUNE = 1 (default: time shapes the table lacks on first use), "table" (never time: static heuristic for missing
    shapes) or 0 (static heuristic only).  Results are bit-identical whatever tile is chosen (tests/test_gpu_kernels.py)."""
    if getattr(load_gemm_table, "done", False):
        return
    load_gemm_table.done = True
    mode = os.environ.get("TCL_AUTOTUNE", "1")
    if mode == "0":
        L.tcl_gemm_autotune(0)
        return
    path = os.environ.get("TCL_GEMM_TABLE", GEMM_TABLE)
    if os.path.exists(path):
        try:                    # (the ctypes binding raises on a non-zero return code: an unversioned / foreign table is TCL_EINVAL)
            L.tcl_gemm_tune_load(path)
        except RuntimeError:
            import warnings
            warnings.warn(f"GEMM tile table {path} was written for other kernels (no / other version line): ignored, shapes are timed on first use")
    L.tcl_gemm_autotune(2 if mode == "table" else 1)


class Ops:
    """Thin typed wrappers around the C ABI (allocation via torch's caching allocator, launches on the current stream)."""
    _splitk_ws = {}

    def __init__(self, dev):
        self.dev = dev
        self.L = lib()
        load_gemm_table(self.L)
        self._gn_ws = {}
        ws = Ops._splitk_ws.get(str(dev))
        if ws is None:      # one split-K scratch per device, registered with the library (single compute stream)
            ws = Ops._splitk_ws[str(dev)] = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
            self.L.tcl_set_workspace(ws, ws.numel())

    def empty(self, *shape, dtype=H16):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def gemm(self, a, w, bias=None, resid=None, act=0, out=None, M=None, lda=None, N=None, K=None, ldw=None, ldc=None):
        if N is None:
            N, K = w.shape
        M = M if M is not None else a.numel() // K
        No = N // 2 if act == 2 else N                      # act 2 = fused GEGLU: output is [M, N/2]
        c = out if out is not None else self.empty(M, No)
        self.L.tcl_gemm_f16(a, w, bias if bias is not None else 0, resid if resid is not None else 0, c, M, N, K,
                            lda or K, ldw or K, ldc or No, N, act, stream())
        return c

    def conv3x3(self, x, B, Hh, Ww, cin, w, bias, resid=None, stride=1, pad=1, up=None):
        cout = w.shape[0]
        Hu, Wu = up if up else (Hh, Ww)
        Ho = (Hu + 2 - 3) // stride + 1 if pad else (Hu + 1 - 3) // stride + 1
        Wo = (Wu + 2 - 3) // stride + 1 if pad else (Wu + 1 - 3) // stride + 1
        y = self.empty(B * Ho * Wo, cout)
        self.L.tcl_conv3x3_f16(x, w, bias if bias is not None else 0, resid if resid is not None else 0, y, B, Hh, Ww, cin, cout,
                               stride, pad, up[0] if up else 0, up[1] if up else 0, 0, stream())
        return y, Ho, Wo

    def groupnorm(self, x1, c1, gamma, beta, B, HW, eps, silu, x2=None, c2=0, groups=32, raw=False):
        """raw=True: also return the un-normalised concat [x1 | x2] (written by the same pass) -> (y, yraw)."""
        key = (B, c1 + c2)
        ws = self._gn_ws.get(key)
        if ws is None:
            ws = self._gn_ws[key] = torch.zeros(self.L.tcl_groupnorm_workspace_bytes(B, c1 + c2), dtype=torch.uint8, device=self.dev)   # zeroed once
        y = self.empty(B * HW, c1 + c2)
        if raw:
            yr = self.empty(B * HW, c1 + c2)
            self.L.tcl_groupnorm_concat_f16(x1, c1, x2 if x2 is not None else 0, c2, gamma, beta, y, yr, B, HW, groups, eps, int(silu), ws, stream())
            return y, yr
        self.L.tcl_groupnorm_f16(x1, c1, x2 if x2 is not None else 0, c2, gamma, beta, y, B, HW, groups, eps, int(silu), ws, stream())
        return y

    def layernorm(self, x, gamma, beta, rows, C, metric=False):
        y = self.empty(rows, C)
        if metric:          # norm1 of a VidToMe-patched block: the matching metric y / |y| leaves the same kernel
            m = self.empty(rows, C)
            self.L.tcl_layernorm_metric_f16(x, gamma, beta, y, m, rows, C, 1e-5, stream())
            return y, m
        self.L.tcl_layernorm_f16(x, gamma, beta, y, rows, C, 1e-5, stream())
        return y

    def ln_gemm(self, x, gamma, beta, w, bias=None, act=0):
        """act(LayerNorm(x) @ w.T + bias) in one kernel (K = 320: csrc/linstrip.hip); the normalised activations are never written."""
        N, K = w.shape
        M = x.numel() // K
        c = self.empty(M, N // 2 if act == 2 else N)
        self.L.tcl_ln_gemm_f16(x, gamma, beta, 1e-5, w, bias if bias is not None else 0, 0, c, M, N, K, K, K, c.shape[1], N, act, stream())
        return c

    def attention(self, q, ldq, qbs, k, ldk, kbs, v, ldv, vbs, B, Hh, Tq, Tk, d, kv_div=1, ws_kv=None, pack_kv=1, pair=False, packed=None):
        """packed: (ws_q, ws_kv) already filled by attention_pack (on whatever stream; the caller orders the two)."""
        o = self.empty(B * Tq, Hh * d)
        if packed is not None:
            wq, ws_kv, flags = packed[0], packed[1], 4
        else:
            wq = torch.empty(self.L.tcl_attention_q_bytes(B, Hh, Tq, d), dtype=torch.uint8, device=self.dev)
            if ws_kv is None:
                ws_kv = torch.empty(self.L.tcl_attention_kv_bytes(B // kv_div, Hh, Tk, d), dtype=torch.uint8, device=self.dev)
            flags = pack_kv
        self.L.tcl_attention_f16(q, ldq, qbs, k if k is not None else 0, ldk, kbs, v if v is not None else 0, ldv, vbs, o, Hh * d,
                                 Tq * Hh * d, B, Hh, Tq, Tk, d, d ** -0.5, kv_div, flags | (2 if pair else 0), wq, ws_kv, stream())
        return o

    def attention_pack(self, q, ldq, qbs, k, ldk, kbs, v, ldv, vbs, B, Hh, Tq, Tk, d):
        """Q / K / V^T panels of one attention call, written on the current stream -> (ws_q, ws_kv) for attention(..., packed=)."""
        wq = torch.empty(self.L.tcl_attention_q_bytes(B, Hh, Tq, d), dtype=torch.uint8, device=self.dev)
        wkv = torch.empty(self.L.tcl_attention_kv_bytes(B, Hh, Tk, d), dtype=torch.uint8, device=self.dev)
        self.L.tcl_attention_pack_f16(q, ldq, qbs, k, ldk, kbs, v, ldv, vbs, B, Hh, Tq, Tk, d, d ** -0.5, 1, 1, wq, wkv, stream())
        return wq, wkv


class UNetEngine:
    """unet(sample[2F,4,h',w'], t, encoder_hidden_states=[2F,L,768], cross_attention_kwargs={'concat_conds': ...}).sample
    is served by `forward_nhwc` (NHWC in/out) so that the pack/unpack kernels can feed it without layout round trips."""

    in_channels = 4   # unet.config.in_channels stays 4 (generate.py:174)

    def __init__(self, state_dict, device, vidtome=None):
        self.dev = torch.device(device)
        self.ops = Ops(self.dev)
        self.L = lib()
        self.tome = vidtome if vidtome is not None else VidToMe(self.dev)
        sd = state_dict
        exp = sd15.unet_param_shapes()
        missing = [k for k in exp if k not in sd]
        if missing:
            raise KeyError(f"UNet state dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        d = self.dev
        w = {}
        ci = sd["conv_in.weight"]
        wi = torch.zeros(320, 128)
        wi[:, :9 * ci.shape[1]] = ci.permute(0, 2, 3, 1).reshape(320, -1)
        w["conv_in"] = (_dev(wi, d), _dev(sd["conv_in.bias"], d), ci.shape[1])
        w["t1"] = (_dev(sd["time_embedding.linear_1.weight"], d), _dev(sd["time_embedding.linear_1.bias"], d))
        w["t2"] = (_dev(sd["time_embedding.linear_2.weight"], d), _dev(sd["time_embedding.linear_2.bias"], d))
        self.res, self.tfm = {}, {}
        for k in exp:
            if k.endswith("norm1.weight") and ".resnets." in k:
                self.res[k[:-len("norm1.weight")]] = None
            if k.endswith("proj_in.weight"):
                self.tfm[k[:-len("proj_in.weight")]] = None
        for p in self.res:
            r = dict(n1=(_dev(sd[p + "norm1.weight"], d), _dev(sd[p + "norm1.bias"], d)),
                     c1=_conv_w(sd[p + "conv1.weight"], d), b1=_dev(sd[p + "conv1.bias"], d),
                     tw=_dev(sd[p + "time_emb_proj.weight"], d), tb=_dev(sd[p + "time_emb_proj.bias"], d),
                     n2=(_dev(sd[p + "norm2.weight"], d), _dev(sd[p + "norm2.bias"], d)),
                     c2=_conv_w(sd[p + "conv2.weight"], d), b2=_dev(sd[p + "conv2.bias"], d),
                     cin=sd[p + "conv1.weight"].shape[1], cout=sd[p + "conv1.weight"].shape[0])
            if p + "conv_shortcut.weight" in sd:
                r["sc"] = (_dev(sd[p + "conv_shortcut.weight"].flatten(1), d), _dev(sd[p + "conv_shortcut.bias"], d))
            self.res[p] = r
        for p in self.tfm:
            t = p + "transformer_blocks.0."
            c = sd[p + "proj_in.weight"].shape[0]
            self.tfm[p] = dict(
                c=c, gn=(_dev(sd[p + "norm.weight"], d), _dev(sd[p + "norm.bias"], d)),
                pin=(_dev(sd[p + "proj_in.weight"].flatten(1), d), _dev(sd[p + "proj_in.bias"], d)),
                pout=(_dev(sd[p + "proj_out.weight"].flatten(1), d), _dev(sd[p + "proj_out.bias"], d)),
                ln=[(_dev(sd[t + f"norm{i}.weight"], d), _dev(sd[t + f"norm{i}.bias"], d)) for i in (1, 2, 3)],
                qkv=_dev(torch.cat([sd[t + "attn1.to_q.weight"], sd[t + "attn1.to_k.weight"], sd[t + "attn1.to_v.weight"]]), d),
                o1=(_dev(sd[t + "attn1.to_out.0.weight"], d), _dev(sd[t + "attn1.to_out.0.bias"], d)),
                q2=_dev(sd[t + "attn2.to_q.weight"], d),
                kv2=_dev(torch.cat([sd[t + "attn2.to_k.weight"], sd[t + "attn2.to_v.weight"]]), d),
                o2=(_dev(sd[t + "attn2.to_out.0.weight"], d), _dev(sd[t + "attn2.to_out.0.bias"], d)),
                ff1=(_dev(_geglu_rows(sd[t + "ff.net.0.proj.weight"]), d), _dev(_geglu_rows(sd[t + "ff.net.0.proj.bias"]), d)),
                ff2=(_dev(sd[t + "ff.net.2.weight"], d), _dev(sd[t + "ff.net.2.bias"], d)),
                text_kv={})
        for i in range(3):
            w[f"down{i}"] = (_conv_w(sd[f"down_blocks.{i}.downsamplers.0.conv.weight"], d), _dev(sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], d))
            w[f"up{i}"] = (_conv_w(sd[f"up_blocks.{i}.upsamplers.0.conv.weight"], d), _dev(sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], d))
        w["norm_out"] = (_dev(sd["conv_norm_out.weight"], d), _dev(sd["conv_norm_out.bias"], d))
        w["conv_out"] = (_conv_w(sd["conv_out.weight"], d), _dev(sd["conv_out.bias"], d))
        self.w = w
        self._temb_cache = (None, None)
        self.flops = 0.0          # algorithmic FLOPs (2*MACs) of the reference's computation, accumulated per forward for the roofline figure
        self.flops_executed = 0.0  # FLOPs that actually ran (== flops without the CFG-pair de-duplication)
        self.count_flops = False

    # ------------------------------------------------------------------ time embedding (once per timestep)
    def _temb(self, t):
        t = float(t)
        if self._temb_cache[0] == t:
            return self._temb_cache[1]
        o, L = self.ops, self.L
        e0 = o.empty(320); L.tcl_timestep_embed_f16(t, 320, e0, stream())
        e1 = o.empty(1280); L.tcl_gemv_f16(self.w["t1"][0], e0, self.w["t1"][1], 0, e1, 1280, 320, 0, 1, stream())
        emb = o.empty(1280); L.tcl_gemv_f16(self.w["t2"][0], e1, self.w["t2"][1], 0, emb, 1280, 1280, 0, 0, stream())
        proj = {}
        for p, r in self.res.items():   # conv1 bias + time_emb_proj(silu(emb)) folded into one per-channel vector
            v = o.empty(r["cout"])
            L.tcl_gemv_f16(r["tw"], emb, r["tb"], r["b1"], v, r["cout"], 1280, 1, 0, stream())
            proj[p] = v
        self._temb_cache = (t, proj)
        return proj

    # ------------------------------------------------------------------ blocks
    def _resblock(self, p, x, B, Hh, Ww, tproj, skip=None, cskip=0, rep=1):
        """rep: how many identical copies of these B samples the reference computes (FLOP accounting only)."""
        o, r = self.ops, self.res[p]
        HW = Hh * Ww
        cx = r["cin"] - cskip
        fuse_cat = skip is not None and "sc" in r and os.environ.get("TCL_GN_CONCAT", "1") != "0"
        hn = o.groupnorm(x, cx, *r["n1"], B, HW, 1e-5, True, x2=skip, c2=cskip, raw=fuse_cat)
        if fuse_cat:            # norm1's apply pass reads x and skip anyway: it writes the raw concat for the 1x1 shortcut as well (same bits as a concat pass)
            hn, xc = hn
        h1, _, _ = o.conv3x3(hn, B, Hh, Ww, r["cin"], r["c1"], tproj[p])
        hn2 = o.groupnorm(h1, r["cout"], *r["n2"], B, HW, 1e-5, True)
        if "sc" in r:
            if fuse_cat:
                pass
            elif skip is not None:
                xc = o.empty(B * HW, r["cin"])
                self.L.tcl_concat_channels_f16(x, cx, skip, cskip, xc, B * HW, stream())
            else:
                xc = x
            xs = o.gemm(xc, r["sc"][0], r["sc"][1])
            self._fl(rep * 2.0 * B * HW * r["cin"] * r["cout"], 2.0 * B * HW * r["cin"] * r["cout"])
        else:
            xs = x
        out, _, _ = o.conv3x3(hn2, B, Hh, Ww, r["cout"], r["c2"], r["b2"], resid=xs)
        self._fl(rep * 2.0 * B * HW * 9 * (r["cin"] + r["cout"]) * r["cout"], 2.0 * B * HW * 9 * (r["cin"] + r["cout"]) * r["cout"])
        return out

    def _fl(self, f, executed=None):
        """f: algorithmic FLOPs of the reference's computation; executed: what ran here (less where the identical CFG halves are computed once)."""
        if self.count_flops:
            self.flops += f
            self.flops_executed += f if executed is None else executed

    def _text_kv(self, blk, text):
        key = id(text)
        hit = blk["text_kv"].get(key)
        if hit is None:
            o = self.ops
            Bt, Lt, _ = text.shape
            c = blk["c"]
            kv = o.gemm(text, blk["kv2"])                        # [Bt*L, 2C]: K | V
            ws = torch.empty(self.L.tcl_attention_kv_bytes(Bt, sd15.HEADS, Lt, c // sd15.HEADS), dtype=torch.uint8, device=self.dev)
            hit = blk["text_kv"][key] = dict(kv=kv, ws=ws, packed=False, L=Lt, text=text)
        return hit

    def _q_panel(self, B, Hh, Tq, d):
        """A zero-initialised query-panel workspace per shape, reused by every block (stream order serialises them): tcl_ln_gemm_qpanel_f16 writes the
        rows t < Tq of every (sample, head) panel and never touches the padding rows, which therefore stay zero."""
        key = (B, Hh, Tq, d)
        cache = self.__dict__.setdefault("_qpanels", {})
        if key not in cache:
            if len(cache) > 4:
                cache.clear()
            cache[key] = torch.zeros(self.L.tcl_attention_q_bytes(B, Hh, Tq, d), dtype=torch.uint8, device=self.dev)
        return cache[key]

    def panel_cache_cap(self):
        """Byte bound of the persistent attention-panel cache: 8 % of the device's memory, at most 24 GiB (the 288 GB part: 23 GiB; ADVICE r5: the cap
        was a constant the token budget of Generator.__init__ did not know about -- that budget now subtracts this figure from `free`)."""
        cap = getattr(self, "_panel_cap", None)
        if cap is None:
            total = torch.cuda.get_device_properties(self.dev).total_memory if self.dev.type == "cuda" and torch.cuda.is_available() else (288 << 30)
            cap = self._panel_cap = int(min(24 << 30, 0.08 * total))
        return cap

    def _attn_panels(self, ne, Hh, T, d):
        """Zero-initialised Q and K / V^T panel workspaces per (entries, heads, merged length, head_dim), reused by every chunk and block that meets the shape
        again (all on the main stream: stream order serialises writer and readers).  tcl_gemm_qkv_panels_f16 never writes the panels' padding, which therefore
        stays zero.  The merged lengths of a clip are a handful of values (chunk lengths 1..4 x which side of the global merge is src); the cache is bounded."""
        key = (ne, Hh, T, d)
        cache = self.__dict__.setdefault("_panel_cache", {})
        hit = cache.pop(key, None)
        if hit is None:
            nq, nkv = self.L.tcl_attention_q_bytes(ne, Hh, T, d), self.L.tcl_attention_kv_bytes(ne, Hh, T, d)
            while cache and sum(a.numel() + b.numel() for a, b in cache.values()) + nq + nkv > self.panel_cache_cap():
                cache.pop(next(iter(cache)))                      # oldest first (dict order = insertion / last use)
            try:
                hit = (torch.zeros(nq, dtype=torch.uint8, device=self.dev), torch.zeros(nkv, dtype=torch.uint8, device=self.dev))
            except torch.OutOfMemoryError:                        # a shared / smaller device: give the cached panels back and try once more
                cache.clear()
                torch.cuda.empty_cache()
                hit = (torch.zeros(nq, dtype=torch.uint8, device=self.dev), torch.zeros(nkv, dtype=torch.uint8, device=self.dev))
        cache[key] = hit
        return hit

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            prio = os.environ.get("TCL_SIDE_PRIO", "")
            if prio == "low":       # experiment: lowest HIP stream priority for the matching chain (torch only hands out default / high ones)
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                lo, hi = ctypes.c_int(0), ctypes.c_int(0)
                hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
                h = ctypes.c_void_p()
                rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 0, lo.value)
                assert rc == 0, rc
                print(f"[unet] side stream priority {lo.value} (range least {lo.value} .. greatest {hi.value})", flush=True)
                self._side = torch.cuda.ExternalStream(h.value, device=self.dev)
            else:
                self._side = torch.cuda.Stream(device=self.dev, priority=-1 if prio == "high" else 0)
        return self._side

    def _transformer(self, p, x, B, Fs, Hh, Ww, text, pair_half=False, skew=False, chunks=None):
        """(A generator: use with `yield from`.  skew=False never yields.  skew=True yields "M" before the block's matching chain
        is issued, "A" before its attn1 launches and "F" behind them -- the points at which forward_pair switches between its two groups.)
        x [B*N, C] with B = 2*sum(Fs) samples: the unconditional samples of all chunks (chunk order), then the conditional ones.
        Everything is batched over the chunks except attn1 of the merging levels, which runs chunk by chunk in the reference order
        because each chunk's merge uses -- and updates -- this block's global-token bank (patch.py:59-82).
        pair_half: x holds only the FIRST half of the B samples (see forward_many: the two classifier-free-guidance halves are identical up to
        the first text cross-attention); proj_in, norm1, the VidToMe merge and attn1 run on that half, the result is duplicated before attn2."""
        o, L, blk = self.ops, self.L, self.tfm[p]
        C, N, Hd = blk["c"], Hh * Ww, sd15.HEADS
        d = C // Hd
        ne = 1 if pair_half else 2                               # batch entries physically present through attn1
        Bx = B // 2 * ne
        M = Bx * N
        Ftot = B // 2
        hn = o.groupnorm(x, C, *blk["gn"], Bx, N, 1e-6, False)
        h = o.gemm(hn, blk["pin"][0], blk["pin"][1])
        self._fl(2.0 * B * N * C * C * 2, 2.0 * M * C * C * 2)
        # ---- attn1 over VidToMe-merged tokens (patch.py:161-179)
        merging = self.tome.merges(N) and os.environ.get("TCL_LN_METRIC", "1") != "0"
        n1, m1 = o.layernorm(h, *blk["ln"][0], M, C, metric=True) if merging else (o.layernorm(h, *blk["ln"][0], M, C), None)
        if not self.tome.merges(N):                             # downsample > max_downsample: per-frame attention
            qkv = o.gemm(n1, blk["qkv"])
            a = o.attention(qkv, 3 * C, N * 3 * C, qkv[:, C:], 3 * C, N * 3 * C, qkv[:, 2 * C:], 3 * C, N * 3 * C, Bx, Hd, N, N, d, pair=pair_half)
            h = o.gemm(a, blk["o1"][0], blk["o1"][1], resid=h)
            self._fl(2.0 * B * N * C * C * 4 + 4.0 * B * N * N * C, 2.0 * M * C * C * 4 + 4.0 * Bx * N * N * C)
        else:
            off, xbs = 0, Ftot * N * C                          # a chunk's conditional rows sit xbs elements after its unconditional ones
            # The matching chain (bank order, dozens of small launches per chunk) runs on a side stream, ahead of the attention of the
            # chunks already matched: only merge(c) -> merge(c+1) and merge(c) -> attention(c) are real dependencies, so the small
            # kernels of chunk c+1 fill the tails of chunk c's QKV GEMM / flash launches instead of serialising with them.
            qkv_panel = d in (40, 80) and os.environ.get("TCL_QKV_PANEL", "1") != "0"

            def attend(F, off, merged, unm, T, qkv, packed):
                if qkv is None and qkv_panel:
                    # the QKV projection writes the attention panels itself (gemm.hip QP epilogue): no [ne*T, 3C] tensor, no pack launch; a lazily merged
                    # sequence (block, entry stride, merge map) is gathered by the GEMM's operand load
                    wq, wkv = self._attn_panels(ne, Hd, T, d)
                    src, sbs, idx = merged if isinstance(merged, tuple) else (merged, T * C, 0)
                    L.tcl_gemm_qkv_panels_f16(src, sbs, idx, blk["qkv"], ne, T, Hd, d, C, C, C, d ** -0.5, wq, wkv, stream())
                    a = o.attention(wq, 3 * C, T * 3 * C, None, 0, 0, None, 0, 0, ne, Hd, T, T, d, pair=pair_half, packed=(wq, wkv))
                else:
                    if qkv is None:
                        qkv = o.gemm(merged, blk["qkv"], M=ne * T)
                    a = o.attention(qkv, 3 * C, T * 3 * C, qkv[:, C:], 3 * C, T * 3 * C, qkv[:, 2 * C:], 3 * C, T * 3 * C, ne, Hd, T, T, d, pair=pair_half,
                                    packed=packed)
                y = o.gemm(a, blk["o1"][0], blk["o1"][1], M=ne * T)
                self.tome.unmerge_add(h[off * N:], xbs, y, T, unm, F * N, C, ne)  # u_a(...) + x (patch.py:178-179)
                self._fl(2.0 * 2 * T * C * C * 4 + 4.0 * 2 * T * T * C, 2.0 * ne * T * C * C * 4 + 4.0 * ne * T * T * C)

            if skew:
                yield "M"
            main = torch.cuda.current_stream()
            # (Always on the side stream, single-chunk passes included: the banks then live in that stream's allocator pool for good.)
            two = os.environ.get("TCL_TOME_STREAM", "1") != "0"
            side = self._side_stream() if two else main
            if two:
                side.wait_stream(main)                          # n1 is ready (skew: and the other group's attn1, issued just before, is done)
            # TCL_QKV_SIDE=1: the chunk's QKV projection and panel packing ride on the side stream too (they only depend on the merge, and the main
            # stream is the critical path of the pass).  Measured: -0.7 % denoise time on a same-box A/B, +0.4 % frames/s on the full clip, but the
            # flash kernel's in-pass rate drops 3.5 % (more work shares the matrix pipes with it): off by default.
            qkv_side = two and os.environ.get("TCL_QKV_SIDE", "0") != "0"
            pend = []

            def hand_over(merged, unm, qkv, packed, ev):
                if two:
                    main.wait_event(ev)
                    held = ((qkv,) + packed) if qkv_side else ((merged[0], merged[2]) if isinstance(merged, tuple) else (merged,))
                    for tns in held:                                               # allocated on the side stream's pool, read on the main stream
                        if tns is not None:
                            tns.record_stream(main)
                    if unm is not None:
                        unm.record_stream(main)

            for ci, F in enumerate(Fs):
                self.tome.select_chunk(ci, chunks)
                with torch.cuda.stream(side):
                    merged, unm, T = self.tome.compute_merge(p, n1[off * N:], F, N, C, xbs=xbs, metric=m1[off * N:] if m1 is not None else None, ne=ne,
                                                             lazy_merged=qkv_panel and not qkv_side and os.environ.get("TCL_MERGE_LAZY", "1") != "0")     # merged [ne, T, C]
                    qkv = packed = None
                    if qkv_side:
                        qkv = o.gemm(merged, blk["qkv"], M=ne * T)
                        packed = o.attention_pack(qkv, 3 * C, T * 3 * C, qkv[:, C:], 3 * C, T * 3 * C, qkv[:, 2 * C:], 3 * C, T * 3 * C, ne, Hd, T, T, d)
                ev = None
                if two:
                    ev = torch.cuda.Event()
                    ev.record(side)
                if skew:                                        # the whole chain of the group first; its attn1 when forward_pair comes back
                    pend.append((F, off, merged, unm, T, qkv, packed, ev))
                else:
                    hand_over(merged, unm, qkv, packed, ev)
                    attend(F, off, merged, unm, T, qkv, packed)
                off += F
            if skew:
                yield "A"
                for F, off_c, merged, unm, T, qkv, packed, ev in pend:
                    hand_over(merged, unm, qkv, packed, ev)
                    attend(F, off_c, merged, unm, T, qkv, packed)
                del pend
                yield "F"
            # (running the attention of alternate chunks on a second stream as well was measured: no further gain)
        if pair_half:                                           # from here on the halves differ (text): both exist
            h, x = torch.cat([h, h]), torch.cat([x, x])
            M = B * N
        F = Ftot
        # ---- attn2: text cross-attention on the full tokens
        fuse_ln = C == 320 and os.environ.get("TCL_LN_GEMM", "1") != "0"      # norm2 / norm3 ride in the consumer's operand load (strip-resident Linear)
        tk = self._text_kv(blk, text)
        Lt = tk["L"]
        if fuse_ln and tk["packed"] and os.environ.get("TCL_QPANEL", "1") != "0":
            # norm2 -> to_q straight into the attention kernel's query panel (linstrip.hip Q-panel epilogue): no [M, C] output, no pack pass
            wq = self._q_panel(B, Hd, N, d)
            L.tcl_ln_gemm_qpanel_f16(h, *blk["ln"][1], 1e-5, blk["q2"], M, Hd, d, N, C, C, d ** -0.5, wq, stream())
            a = o.attention(wq, C, N * C, None, 0, 0, None, 0, 0, B, Hd, N, Lt, d, kv_div=F, packed=(wq, tk["ws"]))
        else:
            q = o.ln_gemm(h, *blk["ln"][1], blk["q2"]) if fuse_ln else o.gemm(o.layernorm(h, *blk["ln"][1], M, C), blk["q2"])
            a = o.attention(q, C, N * C, tk["kv"], 2 * C, Lt * 2 * C, tk["kv"][:, C:], 2 * C, Lt * 2 * C, B, Hd, N, Lt, d,
                            kv_div=F, ws_kv=tk["ws"], pack_kv=0 if tk["packed"] else 1)
        tk["packed"] = True
        h = o.gemm(a, blk["o2"][0], blk["o2"][1], resid=h)
        self._fl(2.0 * M * C * C * 2 + 4.0 * M * Lt * C)
        # ---- GEGLU feed-forward
        if fuse_ln:
            f2 = o.ln_gemm(h, *blk["ln"][2], blk["ff1"][0], blk["ff1"][1], act=2)
        else:
            f2 = o.gemm(o.layernorm(h, *blk["ln"][2], M, C), blk["ff1"][0], blk["ff1"][1], act=2)   # Linear(C -> 8C) + GEGLU fused in the GEMM epilogue -> [M, 4C]
        h = o.gemm(f2, blk["ff2"][0], blk["ff2"][1], resid=h)
        self._fl(2.0 * M * C * C * 12)
        return o.gemm(h, blk["pout"][0], blk["pout"][1], resid=x)

    # ------------------------------------------------------------------ forward
    # All chunks of a denoising step go through the UNet in ONE pass, stacked on the batch axis (unconditional samples of every
    # chunk first, then the conditional ones).  The only chunk-to-chunk dependency of the reference loop (generate.py:220-224) is
    # the global-token bank of the merging transformer blocks (levels 0 and 1; patch.py:59-82): chunk c's attn1 in block b needs
    # the bank block b was left with by chunk c-1.  Running block-major (all chunks through block b, then block b+1) keeps both
    # orders -- data flow per chunk, bank order per block -- so inside a block only attn1 loops over the chunks; ResNet blocks,
    # norms, cross-attention, feed-forward and the non-merging levels see 8-31x more rows per GEMM and one launch instead of one
    # per chunk, and every weight matrix is streamed once per step.
    def forward_many(self, x_in, Fs, Hh, Ww, t, text, cfg_pair=False):
        """One block-major pass; see _forward_gen for the arguments."""
        g = self._forward_gen(x_in, Fs, Hh, Ww, t, text, cfg_pair)
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value

    # Two groups of chunks through the UNet half a transformer block apart (TCL_SKEW=1, generate.py).  `tools/micro/corun.py` (round 4,
    # profiles/r4_corun.txt): the matching chain beside the head_dim-40 flash kernel gains NOTHING over running the two one after the other
    # (1.00x), beside a store-bound Linear or a 3x3 conv it gains 6-11 %.  In one block-major pass the chain of a block can only run beside that
    # block's own attn1 -- nothing else of the pass is ready.  With two groups A (the first chunks) and B there is: A's chain of block b is issued
    # behind B's attn1 of block b-1 and runs beside B's cross-attention / feed-forward / ResNet blocks; B's chain of block b behind A's attn1
    # of block b, beside A's.  Order on the main stream per merging block:  F(B,b-1) P(B,b) | A(A,b) | F(A,b) P(A,b+1) | A(B,b)  (P = everything
    # up to norm1, A = attn1, F = the rest), on the side stream  M(A,b) | M(B,b): bank order is A's chunks, then B's, in every block, as in the
    # reference loop -- and every kernel is batch-row independent, so the result is the one of a single pass, bit for bit.
    def forward_pair(self, xa, Fsa, xb, Fsb, Hh, Ww, t, text, cfg_pair=False):
        ca = list(self.tome.begin_step(Fsa, (Hh, Ww)))          # the draws in the reference's chunk order: A's chunks, then B's
        cb = list(self.tome.begin_step(Fsb, (Hh, Ww)))
        gens = {"a": self._forward_gen(xa, Fsa, Hh, Ww, t, text, cfg_pair, skew=True, chunks=ca),
                "b": self._forward_gen(xb, Fsb, Hh, Ww, t, text, cfg_pair, skew=True, chunks=cb)}
        res = {}

        def step(k):
            if k in res:
                return
            try:
                next(gens[k])
            except StopIteration as e:
                res[k] = e.value

        step("a"); step("a")                                    # A runs two segments ahead: [P0] [M0]
        while len(res) < 2:
            step("b"); step("a")
        return res["a"], res["b"]

    def _forward_gen(self, x_in, Fs, Hh, Ww, t, text, cfg_pair=False, skew=False, chunks=None):
        """x_in [2*Ftot, Hh, Ww, 8] f16 (latents | concat_conds; Ftot = sum(Fs) samples in chunk order, twice: uncond, cond);
        text [2, L, 768] f16 (uncond, cond).  Fs: chunk lengths in the reference's chunk order.  -> eps [2*Ftot, Hh, Ww, 4] f16.
        cfg_pair: the caller GUARANTEES x_in[Ftot:] == x_in[:Ftot] (the classifier-free-guidance pair of generate.py:342-347: `torch.cat([latents] * 2)`
        and the same concat_conds, written twice by tcl_pack_latents_f16).  The two halves then differ only through the text, which first enters in
        attn2 of the first transformer block: conv_in, the first ResNet block and proj_in / norm1 / VidToMe merge / attn1 of that block run ONCE and
        are duplicated -- the same bits as computing them twice (every kernel is deterministic and batch-row independent; with equal scores in both
        entries the matching picks the first: `test_cfg_pair_dedup_bit_identical`).  TCL_CFG_DEDUP=0 computes both halves."""
        o, L, w = self.ops, self.L, self.w
        Ftot = sum(Fs)
        B = 2 * Ftot
        half = bool(cfg_pair) and os.environ.get("TCL_CFG_DEDUP", "1") != "0"
        Bh = Ftot if half else B                                # samples through the text-free prefix
        tproj = self._temb(t)
        if chunks is None:
            self.tome.begin_step(Fs, (Hh, Ww))                  # every chunk's lock-step draws, in chunk order (patch.py:206-231)
        col = o.empty(Bh * Hh * Ww, 128)
        L.tcl_im2col3x3_small_f16(x_in, col, Bh, Hh, Ww, w["conv_in"][2], 128, stream())
        h = o.gemm(col, w["conv_in"][0], w["conv_in"][1])
        del col
        self._fl(2.0 * B * Hh * Ww * 72 * 320, 2.0 * Bh * Hh * Ww * 72 * 320)
        sizes = [(Hh, Ww)]
        skips = [(torch.cat([h, h]) if half else h, 320)]
        hh, ww, c = Hh, Ww, 320
        for i, co in enumerate(sd15.BLOCK_OUT):
            for j in range(2):
                first = half and i == 0 and j == 0
                h = self._resblock(f"down_blocks.{i}.resnets.{j}.", h, Bh if first else B, hh, ww, tproj, rep=2 if first else 1)
                c = co
                if i < 3:
                    h = yield from self._transformer(f"down_blocks.{i}.attentions.{j}.", h, B, Fs, hh, ww, text, pair_half=first, skew=skew, chunks=chunks)
                skips.append((h, c))
            if i < 3:
                h, hh2, ww2 = o.conv3x3(h, B, hh, ww, c, *w[f"down{i}"], stride=2, pad=1)
                self._fl(2.0 * B * hh2 * ww2 * 9 * c * c)
                hh, ww = hh2, ww2
                sizes.append((hh, ww))
                skips.append((h, c))
        h = self._resblock("mid_block.resnets.0.", h, B, hh, ww, tproj)
        h = yield from self._transformer("mid_block.attentions.0.", h, B, Fs, hh, ww, text, skew=skew, chunks=chunks)
        h = self._resblock("mid_block.resnets.1.", h, B, hh, ww, tproj)
        level = 3
        for i, co in enumerate(sd15.BLOCK_OUT[::-1]):
            for j in range(3):
                sk, cs = skips.pop()
                h = self._resblock(f"up_blocks.{i}.resnets.{j}.", h, B, hh, ww, tproj, skip=sk, cskip=cs)
                del sk
                if i > 0:
                    h = yield from self._transformer(f"up_blocks.{i}.attentions.{j}.", h, B, Fs, hh, ww, text, skew=skew, chunks=chunks)
            if i < 3:
                level -= 1
                h, hh2, ww2 = o.conv3x3(h, B, hh, ww, co, *w[f"up{i}"], up=sizes[level])   # nearest upsample fused in the gather
                self._fl(2.0 * B * hh2 * ww2 * 9 * co * co)
                hh, ww = hh2, ww2
        hn = o.groupnorm(h, 320, *w["norm_out"], B, hh * ww, 1e-5, True)
        eps, _, _ = o.conv3x3(hn, B, hh, ww, 320, *w["conv_out"])
        self._fl(2.0 * B * hh * ww * 9 * 320 * 4)
        self.tome.end_forward()
        return eps

    def forward_nhwc(self, x_in, F, Hh, Ww, t, text):
        """One chunk: x_in [2F, Hh, Ww, 8] f16 -> eps [2F, Hh, Ww, 4] f16."""
        return self.forward_many(x_in, [F], Hh, Ww, t, text)
