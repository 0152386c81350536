// Shared helpers for the gfx950 kernels of tc_light_amd (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define TCL_OK 0
#define TCL_EINVAL 1   // bad argument (null pointer, unsupported shape)
#define TCL_ELAUNCH 2  // hip launch error

#define TCL_CHECK_ARG(cond) do { if (!(cond)) return TCL_EINVAL; } while (0)
#define TCL_LAUNCH_RET() do { return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH; } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// grid for HBM-streaming kernels: cap at 256 CUs x 8 blocks and grid-stride the rest
static inline int stream_grid(long n, int block = 256, int per_thread = 1) {
    long g = (n + (long)block * per_thread - 1) / ((long)block * per_thread);
    return (int)(g < 1 ? 1 : (g > 2048 * 4 ? 2048 * 4 : g));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float* red /* >= 16 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x < 64) {
        r = (lane < nw) ? red[lane] : 0.f;
        r = wave_sum(r);
    }
    return r;
}
