// MemFlowNet correlation lookup for gfx950 (reference: utils/evaluation/memflow/core/Networks/MemFlowNet/corr.py:74-120
// `CorrBlock`, and the windowed formulation of its unused `alt_cuda_corr` extension, correlation_kernel.cu:18-119).
// The reference materialises the all-pairs volume [B*H*W, 1, H, W] (829 MB per frame pair at 1280x720 / 8) plus a 4-level pyramid and
// samples (2r+1)^2 windows from it every GRU iteration.  Pooling and bilinear sampling are linear in fmap2, so here a window is computed
// on demand from the avg-pooled fmap2 pyramid: per pixel and level, dots with the (2r+2)^2 integer neighbours of the centre, then the
// bilinear mix -- O(HW) memory, HBM-bound (the neighbour rows of adjacent pixels overlap and hit in L2).
// One wave per (pixel, level): lanes split the feature dimension, partial dots are transposed through LDS in chunks of 32 points.
#include "common.h"
#include "../../include/tclight_hip.h"

__global__ void k_avgpool2_nhwc(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int D) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D); long p = i / D; const int xo = (int)(p % Wo); p /= Wo; const int yo = (int)(p % Ho); const long b = p / Ho;
        const float* s = x + ((b * H + 2 * yo) * W + 2 * xo) * D + d;
        y[i] = 0.25f * (s[0] + s[D] + s[(long)W * D] + s[(long)W * D + D]);
    }
}

struct CorrLevels { const float* f2[4]; int H[4], W[4]; };

// out[b][(l*n + a)*n + b2] at pixel (y,x): x offset a - r, y offset b2 - r (the reference's meshgrid(dy, dx) order), * 1/sqrt(D)
__global__ __launch_bounds__(256) void k_corr_lookup(const float* __restrict__ f1, CorrLevels lv, const float* __restrict__ coords, float* __restrict__ out,
                                                     int B, int H, int W, int D, int r, long out_cs, long out_ps, long out_bs, float inv_sqrt_d) {
    __shared__ float part[4][32][65];
    __shared__ float dots[4][144];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, l = blockIdx.y, b = blockIdx.z;
    const int pix = blockIdx.x * 4 + wid, P = H * W;
    const bool live = pix < P;
    const int n1 = 2 * r + 2, npts = n1 * n1, n = 2 * r + 1;            // r <= 5 -> npts <= 144
    const int Hl = lv.H[l], Wl = lv.W[l];
    const float* g = lv.f2[l] + (long)b * Hl * Wl * D;
    float cx = 0.f, cy = 0.f;
    if (live) { const float s = 1.f / (float)(1 << l); cx = coords[((long)b * 2 + 0) * P + pix] * s; cy = coords[((long)b * 2 + 1) * P + pix] * s; }
    const float x0f = floorf(cx), y0f = floorf(cy), fx = cx - x0f, fy = cy - y0f;
    const int x0 = (int)x0f - r, y0 = (int)y0f - r;
    float q[8];
    const int nv = D >> 6;                                             // D % 64 == 0, D <= 512
    if (live)
        for (int j = 0; j < nv; ++j) q[j] = f1[((long)b * P + pix) * D + lane + 64 * j];
    for (int c0 = 0; c0 < npts; c0 += 32) {
        for (int pt = 0; pt < 32; ++pt) {
            const int p = c0 + pt;
            float acc = 0.f;
            if (live && p < npts) {
                const int iy = y0 + p / n1, ix = x0 + p % n1;
                if (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) {
                    const float* row = g + ((long)iy * Wl + ix) * D + lane;
                    for (int j = 0; j < nv; ++j) acc += q[j] * row[64 * j];
                }
            }
            part[wid][pt][lane] = acc;
        }
        __syncthreads();
        {   // lane sums half of point (lane & 31)'s 64 partials in a fixed order, halves combined by one exchange
            const int pt = lane & 31, h0 = (lane >> 5) * 32;
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += part[wid][pt][h0 + k];
            s += __shfl_xor(s, 32, 64);
            if (lane < 32 && c0 + pt < npts) dots[wid][c0 + pt] = s;
        }
        __syncthreads();
    }
    if (!live) return;
    float* o = out + (long)b * out_bs + (long)pix * out_ps + (long)l * n * n * out_cs;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int a = idx / n, b2 = idx % n;                            // a: x offset, b2: y offset
        const float* d0 = &dots[wid][b2 * n1 + a];
        const float v = (1.f - fx) * (1.f - fy) * d0[0] + fx * (1.f - fy) * d0[1] + (1.f - fx) * fy * d0[n1] + fx * fy * d0[n1 + 1];
        o[(long)idx * out_cs] = v * inv_sqrt_d;
    }
}

extern "C" {

int tcl_avgpool2_nhwc_f32(const float* x, float* y, int B, int H, int W, int D, hipStream_t st) {
    TCL_CHECK_ARG(x && y && B > 0 && H >= 2 && W >= 2 && D > 0);
    hipLaunchKernelGGL(k_avgpool2_nhwc, dim3(stream_grid((long)B * (H / 2) * (W / 2) * D, 256, 1)), dim3(256), 0, st, x, y, B, H, W, D);
    TCL_LAUNCH_RET();
}

int tcl_corr_lookup_f32(const float* fmap1, const float* const* fmap2_levels, const int* level_h, const int* level_w, int num_levels,
                        const float* coords, float* out, int B, int H, int W, int D, int radius, int out_nchw, hipStream_t st) {
    TCL_CHECK_ARG(fmap1 && fmap2_levels && level_h && level_w && coords && out && B > 0 && H > 0 && W > 0);
    TCL_CHECK_ARG(num_levels >= 1 && num_levels <= 4 && radius >= 1 && radius <= 5 && D % 64 == 0 && D <= 512);
    CorrLevels lv;
    for (int i = 0; i < 4; ++i) { lv.f2[i] = i < num_levels ? fmap2_levels[i] : nullptr; lv.H[i] = i < num_levels ? level_h[i] : 0; lv.W[i] = i < num_levels ? level_w[i] : 0; }
    for (int i = 0; i < num_levels; ++i) TCL_CHECK_ARG(lv.f2[i] && lv.H[i] > 0 && lv.W[i] > 0);
    const int n = 2 * radius + 1, P = H * W, C = num_levels * n * n;
    const long cs = out_nchw ? P : 1, ps = out_nchw ? 1 : C, bs = (long)C * P;
    hipLaunchKernelGGL(k_corr_lookup, dim3(cdiv(P, 4), num_levels, B), dim3(256), 0, st, fmap1, lv, coords, out, B, H, W, D, radius, cs, ps, bs,
                       1.f / sqrtf((float)D));
    TCL_LAUNCH_RET();
}

}  // extern "C"
