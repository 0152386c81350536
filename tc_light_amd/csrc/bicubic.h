// Bicubic (A = -0.75) backward-warp taps shared by path2.hip and producer.hip (grid_sample bicubic / zeros / align_corners).
#pragma once
#include <hip/hip_runtime.h>

// No fma contraction in cubic_w / make_tap (round 5, second pass): hipcc is free to fuse `a * b - floor(a * b)` or `(g - 0.5) * 2 + 1` one way in one
// kernel and another way in the next (it did: after an unrelated rewrite of k_flow_loss its tiled instantiation computed the taps with v_pk_fma_f32 and no
// longer agreed bit for bit with the untiled one, nor with the previous build) -- with the pragma every kernel walks the reference's float sequence literally.
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
#pragma clang fp contract(off)
    const float A = -0.75f;
    float x = t + 1.f;
    w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    x = 1.f - t;
    w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 2.f - t;
    w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
struct Tap { int x0, y0; float wx[4], wy[4]; };
__device__ __forceinline__ Tap make_tap(float fx, float fy, int x, int y, int W, int H) {
    // same float sequence as the reference: normalise to [-1,1] (flow_utils.py:12-13) then un-normalise
#pragma clang fp contract(off)
    float gx = ((fx + (float)x) / (float)(W - 1) - 0.5f) * 2.f;
    float gy = ((fy + (float)y) / (float)(H - 1) - 0.5f) * 2.f;
    float ix = (gx + 1.f) * 0.5f * (float)(W - 1);
    float iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    float fx0 = floorf(ix), fy0 = floorf(iy);
    Tap t; t.x0 = (int)fx0 - 1; t.y0 = (int)fy0 - 1;
    cubic_w(ix - fx0, t.wx); cubic_w(iy - fy0, t.wy);
    return t;
}

