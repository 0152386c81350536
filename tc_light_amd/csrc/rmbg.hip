// BriaRMBG-1.4 (U^2-Net) kernels for gfx950, f32 NCHW (reference briarmbg.py, generate.py:147-167).
// The net is ~0.2 TFLOP per frame, once per video: a VALU direct convolution is enough (no MFMA): every thread owns one output pixel
// and 16 output channels; per input channel it loads the 9 (dilated) taps once and applies 9 x 16 weights that are wave-uniform
// (scalar loads from the tap-major [Cin*9][Cout] weight layout).  Fused: decoder channel concat as two sources, folded BatchNorm
// scale/shift (+ conv bias), ReLU, RSU residual.  Plus max-pool 2x2 (ceil_mode), bilinear resize (align_corners=False) with optional
// input scaling, sigmoid and clamp.
#include "common.h"
#include "../../include/tclight_hip.h"

template <int OCT>
__global__ __launch_bounds__(256) void k_conv3x3_direct(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                                                        const float* __restrict__ wt, const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ resid, float* __restrict__ y, int H, int W, int Ho, int Wo,
                                                        int Cout, int dil, int stride, int relu) {
    const int b = blockIdx.z, oc0 = blockIdx.y * OCT;
    const int p = blockIdx.x * 256 + threadIdx.x, P = Ho * Wo;
    const bool live = p < P;
    const int oy = live ? p / Wo : 0, ox = live ? p - oy * Wo : 0;
    const int iy0 = oy * stride - dil, ix0 = ox * stride - dil;       // padding = dilation (REBNCONV), = 1 for the plain convs
    int off[9]; bool ok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int iy = iy0 + (k / 3) * dil, ix = ix0 + (k % 3) * dil;
        ok[k] = live && iy >= 0 && iy < H && ix >= 0 && ix < W;
        off[k] = ok[k] ? iy * W + ix : 0;
    }
    float acc[OCT];
#pragma unroll
    for (int o = 0; o < OCT; ++o) acc[o] = 0.f;
    const long HW = (long)H * W;
    const int C = C1 + C2;
    for (int ci = 0; ci < C; ++ci) {
        const float* src = ci < C1 ? x1 + ((long)b * C1 + ci) * HW : x2 + ((long)b * C2 + (ci - C1)) * HW;
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = ok[k] ? src[off[k]] : 0.f;
        const float* wr = wt + ((long)ci * 9) * Cout + oc0;             // wave-uniform -> scalar loads
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int o = 0; o < OCT; ++o) acc[o] += v[k] * wr[k * Cout + o];
    }
    if (!live) return;
#pragma unroll
    for (int o = 0; o < OCT; ++o) {
        const int oc = oc0 + o;
        if (oc >= Cout) break;
        float r = acc[o] * scale[oc] + shift[oc];
        if (relu) r = fmaxf(r, 0.f);
        const long idx = ((long)b * Cout + oc) * P + p;
        if (resid) r += resid[idx];
        y[idx] = r;
    }
}

__global__ void k_maxpool2_ceil(const float* __restrict__ x, float* __restrict__ y, int BC, int H, int W) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)BC * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo); long t = i / Wo; const int yo = (int)(t % Ho); const long c = t / Ho;
        const float* s = x + (c * H + 2 * yo) * W + 2 * xo;
        const bool hx = 2 * xo + 1 < W, hy = 2 * yo + 1 < H;
        float m = s[0];
        if (hx) m = fmaxf(m, s[1]);
        if (hy) { m = fmaxf(m, s[W]); if (hx) m = fmaxf(m, s[W + 1]); }
        y[i] = m;
    }
}

// F.interpolate(mode="bilinear", align_corners=False): src = (dst + 0.5) * in/out - 0.5 clamped at 0
__global__ void k_resize_bilinear(const float* __restrict__ x, float* __restrict__ y, int BC, int H, int W, int Ho, int Wo, float mul, int sigmoid,
                                  int clamp01) {
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const long total = (long)BC * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo); long t = i / Wo; const int yo = (int)(t % Ho); const long c = t / Ho;
        const float fy = fmaxf(((float)yo + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)xo + 0.5f) * sx - 0.5f, 0.f);
        const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1), y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* s = x + c * (long)H * W;
        float v = (1.f - ly) * ((1.f - lx) * s[(long)y0 * W + x0] + lx * s[(long)y0 * W + x1]) + ly * ((1.f - lx) * s[(long)y1 * W + x0] + lx * s[(long)y1 * W + x1]);
        v *= mul;
        if (sigmoid) v = 1.f / (1.f + __expf(-v));
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        y[i] = v;
    }
}

extern "C" {

int tcl_conv3x3_direct_f32(const float* x1, int C1, const float* x2, int C2, const float* w_t, const float* scale, const float* shift,
                           const float* resid, float* y, int B, int H, int W, int Cout, int dilation, int stride, int relu, hipStream_t st) {
    TCL_CHECK_ARG(x1 && w_t && scale && shift && y && B > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && dilation >= 1 && (stride == 1 || stride == 2));
    TCL_CHECK_ARG(C2 == 0 || x2);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const dim3 blk(256);
    if (Cout % 16 == 0)
        hipLaunchKernelGGL(k_conv3x3_direct<16>, dim3(cdiv((long)Ho * Wo, 256), Cout / 16, B), blk, 0, st, x1, C1, x2, C2, w_t, scale, shift, resid, y, H, W, Ho, Wo,
                           Cout, dilation, stride, relu);
    else
        hipLaunchKernelGGL(k_conv3x3_direct<1>, dim3(cdiv((long)Ho * Wo, 256), Cout, B), blk, 0, st, x1, C1, x2, C2, w_t, scale, shift, resid, y, H, W, Ho, Wo,
                           Cout, dilation, stride, relu);
    TCL_LAUNCH_RET();
}

int tcl_maxpool2_ceil_f32(const float* x, float* y, int BC, int H, int W, hipStream_t st) {
    TCL_CHECK_ARG(x && y && BC > 0 && H > 0 && W > 0);
    hipLaunchKernelGGL(k_maxpool2_ceil, dim3(stream_grid((long)BC * ((H + 1) / 2) * ((W + 1) / 2), 256, 1)), dim3(256), 0, st, x, y, BC, H, W);
    TCL_LAUNCH_RET();
}

int tcl_resize_bilinear_f32(const float* x, float* y, int BC, int H, int W, int Ho, int Wo, float mul, int sigmoid, int clamp01, hipStream_t st) {
    TCL_CHECK_ARG(x && y && BC > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0);
    hipLaunchKernelGGL(k_resize_bilinear, dim3(stream_grid((long)BC * Ho * Wo, 256, 1)), dim3(256), 0, st, x, y, BC, H, W, Ho, Wo, mul, sigmoid, clamp01);
    TCL_LAUNCH_RET();
}

}  // extern "C"
