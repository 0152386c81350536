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

// epilogue activations: 1 SiLU, 3 ReLU, 4 GELU (erf); 0, 2 (GEGLU, applied on column pairs by the caller) and 5 (GELU applied AFTER the
// residual add, see post_act) leave v unchanged
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v / (1.f + __expf(-v));
    if (act == 3) return fmaxf(v, 0.f);
    if (act == 4) return 0.5f * v * (1.f + erff(v * 0.70710678f));
    return v;
}

__device__ __forceinline__ float post_act(float v, int act) { return act == 5 ? 0.5f * v * (1.f + erff(v * 0.70710678f)) : v; }

int gemm8_dispatch(int cfg, const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                   int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st);
