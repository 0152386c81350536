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

int gemm8_dispatch(int cfg, const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                   int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st);
