// Shared between gemm.hip and gemm8.hip: implicit-GEMM 3x3 convolution descriptor and the 8-wave kernel entry.
#pragma once
#include <hip/hip_runtime.h>

struct ConvP {
    int conv;            // 0: dense A[M][lda]; 1: implicit 3x3
    int Hin, Win, Cin;   // stored input (before optional upsample)
    int Hup, Wup;        // logical input size seen by the conv (== Hin,Win unless nearest-upsampled)
    int Hout, Wout, stride, pad;
    float sy, sx;        // Hin/Hup, Win/Wup (nearest source scale, PyTorch 'nearest' convention)
};

// QKV projection written straight into the flash kernel's panels (round 5; csrc/attn.hip documents the layouts): Qp [ne, H, Tqp, DP] pre-scaled,
// Kp [ne, H, Tkp, KS] (+ ones column `one_col` when >= 0), Vt [ne, H, Tkp / 64, vtile halves]: DPV rows of 72 halves per 64-key tile, keys permuted
// (bits 2 <-> 3 of the in-tile key index), rows 4..11 (mod 16) skewed by 16 positions when `skew`, row d = ones over the valid keys when DPV > d.
// Everything the epilogue does not write (padding rows / columns, rows d + 1 .. DPV - 1) is expected to be ZERO already.
// aidx (may be null): token t of an entry reads row aidx[t] of that entry's source block (the VidToMe merge map: the merged sequence is never gathered into
// a tensor of its own); a_bs: elements between the entries' source blocks.
struct QkvPanel { _Float16 *qp, *kp, *vt; int T, Tqp, Tkp, H, d, DP, KS, DPV, vtile, one_col, skew; float qscale; const int* aidx; long a_bs; };
int gemm_dma_qkv_panels(const _Float16* A, const _Float16* W, int ne, int K, int lda, int ldw, const QkvPanel& qp, hipStream_t st);

// K order of the implicit 3x3 convolution, shared by every GEMM kernel so that they all accumulate in the same order (results are
// bit-identical across tile configurations).  The weights are stored tap-major ([Cout][9][Cin]); the kernels WALK K channel-slice-major
// when Cin % 64 == 0: the 64-wide step q covers channels (q/9)*64.. of tap q%9, so nine consecutive steps re-touch the same input rows
// (one 128-B line per pixel, L1/L2-resident) instead of striding through all Cin channels of one tap.  k0 is the linear step start
// (a multiple of 32); returns the tap and sets c0 (channel offset); the weight column is tap*Cin + c0.
__device__ __forceinline__ int conv_kmap(int k0, int Cin, int& c0) {
    if ((Cin & 63) == 0) { const int q = k0 >> 6, c64 = q / 9; c0 = c64 * 64 + (k0 & 63); return q - c64 * 9; }
    const int tap = k0 / Cin; c0 = k0 - tap * Cin; return tap;
}

// erf-GELU for epilogues whose input and output are f16: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the f16 rounding of the
// result) -- one v_rcp, one v_exp and 9 fma / mul instead of libdevice erff's two-branch evaluation (~28 vector instructions per element;
// the GEGLU epilogue of a K = 320 Linear issues more vector cycles than its ten K steps issue matrix cycles)
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = v * 0.70710678f, ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    const float erfa = fmaf(-(p * t), e, 1.f);                     // erf(|x|)
    return 0.5f * v * (1.f + copysignf(erfa, x));
}

// epilogue activations: 1 SiLU, 3 ReLU, 4 GELU (erf); 0, 2 (GEGLU, applied on column pairs by the caller) and 5 (GELU applied AFTER the
// residual add, see post_act) leave v unchanged
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v / (1.f + __expf(-v));
    if (act == 3) return fmaxf(v, 0.f);
    if (act == 4) return gelu_erf(v);
    return v;
}

__device__ __forceinline__ float post_act(float v, int act) { return act == 5 ? gelu_erf(v) : v; }

int gemm8_dispatch(int cfg, const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                   int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st);
// 8-phase 256-row kernels (gemm8q.hip): cfg 1 = 256 x 256, 2 = 256 x 320
bool gemm8q_ok(int cfg, int M, int N, int K, int lda, int ldw, int ldc, int ldr, bool has_resid, int act, const ConvP& cp);
int gemm8q_dispatch(int cfg, const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                    int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st);
// strip-resident K = 320 Linear (linstrip.hip), cfg 12
bool lin_strip_ok(int M, int N, int K, int lda, int ldw, int ldc, int ldr, bool has_resid, int act, const ConvP& cp);
int lin_strip_dispatch(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                       int lda, int ldw, int ldc, int ldr, int act, hipStream_t st, const _Float16* gamma = nullptr, const _Float16* beta = nullptr,
                       float eps = 0.f);
