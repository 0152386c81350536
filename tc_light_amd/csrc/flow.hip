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
template <typename OT>
__global__ __launch_bounds__(256) void k_corr_lookup(const float* __restrict__ f1, CorrLevels lv, const float* __restrict__ coords, OT* __restrict__ out,
                                                     int B, int H, int W, int D, int r, long out_cs, long out_ps, long out_bs, float inv_sqrt_d,
                                                     const int* __restrict__ tile_flags = nullptr) {
    __shared__ float part[4][32][65];
    __shared__ float dots[4][144];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, l = blockIdx.y, b = blockIdx.z;
    const int pix = blockIdx.x * 4 + wid, P = H * W;
    bool live = pix < P;
    if (tile_flags) {      // round 6: the pass behind k_corr_lookup_tile -- only the pixels of the 8x8 tiles that kernel gave up on (wave-uniform: a wave is one pixel)
        const int tx = (W + 7) >> 3, ty = (H + 7) >> 3;
        const int pc = pix < P ? pix : P - 1;
        if (!tile_flags[l * tx * ty + ((pc / W) >> 3) * tx + ((pc % W) >> 3)]) live = false;
        // (the four waves of a block may disagree; every wave still takes part in nothing block-wide below: part[wid] / dots[wid] are per wave)
    }
    const int n1 = 2 * r + 2, npts = n1 * n1, n = 2 * r + 1;            // r <= 5 -> npts <= 144
    const int Hl = lv.H[l], Wl = lv.W[l];
    const float* g = lv.f2[l] + (long)b * Hl * Wl * D;
    float cx = 0.f, cy = 0.f;
    if (live) { const float s = 1.f / (float)(1 << l); cx = coords[((long)b * 2 + 0) * P + pix] * s; cy = coords[((long)b * 2 + 1) * P + pix] * s; }
    const float x0f = floorf(cx), y0f = floorf(cy), fx = cx - x0f, fy = cy - y0f;
    const int x0 = (int)x0f - r, y0 = (int)y0f - r;
    // lane owns 4 contiguous features of every 256-feature block (float4 loads): D % 256 == 0 -> nv4 blocks, else the scalar layout
    const int nv4 = (D & 255) == 0 ? D >> 8 : 0, nv = D >> 6;             // D % 64 == 0, D <= 512
    float4 q4[2]; float q[8];
    if (live) {
        if (nv4) for (int j = 0; j < nv4; ++j) q4[j] = *(const float4*)(f1 + ((long)b * P + pix) * D + j * 256 + lane * 4);
        else for (int j = 0; j < nv; ++j) q[j] = f1[((long)b * P + pix) * D + lane + 64 * j];
    }
    for (int c0 = 0; c0 < npts; c0 += 32) {
        for (int pt = 0; pt < 32; ++pt) {
            const int p = c0 + pt;
            float acc = 0.f;
            if (live && p < npts) {
                const int iy = y0 + p / n1, ix = x0 + p % n1;
                if (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) {
                    const float* row = g + ((long)iy * Wl + ix) * D;
                    if (nv4) {
                        for (int j = 0; j < nv4; ++j) { const float4 v = *(const float4*)(row + j * 256 + lane * 4); acc += q4[j].x * v.x + q4[j].y * v.y + q4[j].z * v.z + q4[j].w * v.w; }
                    } else {
                        for (int j = 0; j < nv; ++j) acc += q[j] * row[lane + 64 * j];
                    }
                }
            }
            part[wid][pt][lane] = acc;
        }
        __builtin_amdgcn_wave_barrier();                 // part[wid] and dots[wid] are private to this wave: LDS ops of one wave retire in order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {   // lane sums half of point (lane & 31)'s 64 partials in a fixed order, halves combined by one exchange
            const int pt = lane & 31, h0 = (lane >> 5) * 32;
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += part[wid][pt][h0 + k];
            s += __shfl_xor(s, 32, 64);
            if (lane < 32 && c0 + pt < npts) dots[wid][c0 + pt] = s;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (!live) return;
    OT* o = out + (long)b * out_bs + (long)pix * out_ps + (long)l * n * n * out_cs;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int a = idx / n, b2 = idx % n;                            // a: x offset, b2: y offset
        const float* d0 = &dots[wid][b2 * n1 + a];
        const float v = (1.f - fx) * (1.f - fy) * d0[0] + fx * (1.f - fy) * d0[1] + (1.f - fx) * fy * d0[n1] + fx * fy * d0[n1 + 1];
        o[(long)idx * out_cs] = (OT)(v * inv_sqrt_d);
    }
}


// ---- round 6: the lookup with the neighbour rows SHARED by a tile of pixels.
// k_corr_lookup streams 100 rows of 1 KiB per (pixel, level) out of L2 -- 5.9 GB per call at 1280x720 / 8, 660 us, 26 % of a MemFlowNet frame pair
// (profiles/r6_memflow_kernel_stats_before.txt).  The windows of neighbouring pixels overlap almost completely when the flow is smooth: a tile of 8 x 8 pixels
// needs the rows of ONE bounding box of ~(8 / 2^l + 10)^2 points.  A block therefore stages that box, 32 features at a time, in LDS (f16: level 0 is
// the encoder's f16 output as it stands, the pooled levels are rounded once), and every thread takes a quarter of one pixel's 100 window points:
// 25 dot products per thread on v_dot2_f32_f16 with f32 accumulation in a fixed order (deterministic), the bilinear mix of corr.py:95-112 last, same formula and
// order as k_corr_lookup.  166 KB instead of 6.5 MB of L2 reads per tile and level.  A box of more than CT_RMAX points is staged in bands of whole rows (up to
// CT_MAXB); a tile whose box exceeds that (a flow that tears the tile apart) raises its flag and is left to k_corr_lookup, launched behind with the flags as a filter.
#define CT_K 32            // features per LDS slice
#define CT_STR 40          // LDS row stride in halves: 80 B = 5 x 16 B, odd -> 16-byte reads of 64 different rows are conflict-free
#define CT_RMAX 512        // box points a block stages per band (22 x 23); one more all-zero row stands in for points outside the image / the box / the band
#define CT_MAXB 8          // bands per tile before the per-pixel kernel takes over
struct CorrLevelsH { const _Float16* f2[4]; int H[4], W[4]; };
typedef _Float16 ch8 __attribute__((ext_vector_type(8)));
typedef _Float16 ch2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 2) void k_corr_lookup_tile(const _Float16* __restrict__ f1, CorrLevelsH lv, const float* __restrict__ coords,
                                                             _Float16* __restrict__ out, int H, int W, int D, long ld, float inv_sqrt_d,
                                                             int* __restrict__ tile_flags) {
    constexpr int R = 4, N1 = 2 * R + 2, NP = N1 * N1, N = 2 * R + 1, PT = NP / 4;          // 100 window points, 25 per thread
    __shared__ __attribute__((aligned(16))) _Float16 reg[(CT_RMAX + 1) * CT_STR];
    __shared__ float dots[64][NP + 1];
    __shared__ int s_x0[64], s_y0[64], s_box[4];
    __shared__ float s_fx[64], s_fy[64];
    const int tid = threadIdx.x, pixl = tid & 63, part = tid >> 6, l = blockIdx.y;
    const int tx = (W + 7) >> 3, tile = blockIdx.x, P = H * W;
    const int py = (tile / tx) * 8 + (pixl >> 3), px = (tile % tx) * 8 + (pixl & 7);
    const bool live = py < H && px < W;
    const int pix = live ? py * W + px : 0;
    const int Hl = lv.H[l], Wl = lv.W[l];
    const _Float16* __restrict__ g = lv.f2[l];
    if (tid < 64) {
        float cx = 0.f, cy = 0.f;
        if (live) { const float sc = 1.f / (float)(1 << l); cx = coords[pix] * sc; cy = coords[P + pix] * sc; }
        const float x0f = floorf(cx), y0f = floorf(cy);
        // (a coordinate far outside the image must not overflow the int conversion: clamp, the window is outside either way)
        const int x0 = (int)fminf(fmaxf(x0f, -1e6f), 1e6f) - R, y0 = (int)fminf(fmaxf(y0f, -1e6f), 1e6f) - R;
        s_x0[tid] = x0; s_y0[tid] = y0; s_fx[tid] = cx - x0f; s_fy[tid] = cy - y0f;
        int mnx = live ? x0 : 0x3fffffff, mxx = live ? x0 + N1 - 1 : -0x3fffffff, mny = live ? y0 : 0x3fffffff, mxy = live ? y0 + N1 - 1 : -0x3fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, o, 64)); mxx = max(mxx, __shfl_xor(mxx, o, 64));
            mny = min(mny, __shfl_xor(mny, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
        }
        if (tid == 0) {
            mnx = max(mnx, 0); mny = max(mny, 0); mxx = min(mxx, Wl - 1); mxy = min(mxy, Hl - 1);
            s_box[0] = mnx; s_box[1] = mny; s_box[2] = mxx - mnx + 1; s_box[3] = mxy - mny + 1;
        }
    }
    for (int i = tid; i < CT_STR; i += 256) reg[CT_RMAX * CT_STR + i] = (_Float16)0.f;       // the zero row
    __syncthreads();
    const int bx0 = s_box[0], by0 = s_box[1], bw = s_box[2], bh = s_box[3];
    const bool empty = bw <= 0 || bh <= 0;
    // a box of more than CT_RMAX points is taken in BANDS of whole box rows (each band one staging pass per feature slice; a window point contributes in the band
    // that holds its row, the accumulators live across bands): up to CT_MAXB bands -- 4 096 points, a 64 x 64 box -- still cost less than the per-pixel kernel's
    // 100 L2 rows per pixel; beyond that (or a box wider than CT_RMAX) the tile raises its flag and is left to k_corr_lookup.  Block-uniform decisions.
    const int rows_pb = empty ? 1 : max(CT_RMAX / max(bw, 1), 0), nbands = empty ? 0 : (rows_pb > 0 ? (bh + rows_pb - 1) / rows_pb : CT_MAXB + 1);
    if (nbands > CT_MAXB) { if (tid == 0) tile_flags[l * gridDim.x + tile] = 1; return; }
    if (tid == 0) tile_flags[l * gridDim.x + tile] = 0;
    // this thread's 25 window points -> (box row, box column), -1 outside the image (= outside the clipped box)
    int ryx[PT];
    {
        const int x0 = s_x0[pixl], y0 = s_y0[pixl];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int p = part * PT + j, ry = y0 + p / N1 - by0, rx = x0 + p % N1 - bx0;
            ryx[j] = (live && !empty && ry >= 0 && ry < bh && rx >= 0 && rx < bw) ? ((ry << 12) | rx) : -1;      // bw <= CT_RMAX = 512 < 4096
        }
    }
    float acc[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) acc[j] = 0.f;
    const _Float16* __restrict__ q = f1 + (long)pix * D;
    for (int band = 0; band < nbands; ++band) {
        const int y_lo = band * rows_pb, y_hi = min(y_lo + rows_pb, bh), nreg = (y_hi - y_lo) * bw;
        int off[PT];                                       // LDS row offsets (halves) of the window points inside this band; the zero row otherwise
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int ry = ryx[j] >> 12, rx = ryx[j] & 4095;
            off[j] = (ryx[j] >= 0 && ry >= y_lo && ry < y_hi) ? ((ry - y_lo) * bw + rx) * CT_STR : CT_RMAX * CT_STR;
        }
        for (int ks = 0; ks < D; ks += CT_K) {
            __syncthreads();                                  // the previous slice has been read by everybody
            for (int i = tid >> 2; i < nreg; i += 64) {      // 4 threads per row: 4 x 16 B = the row's 32 features of this slice
                const int iy = by0 + y_lo + i / bw, ix = bx0 + i % bw;
                *(ch8*)(reg + i * CT_STR + (tid & 3) * 8) = *(const ch8*)(g + ((long)iy * Wl + ix) * D + ks + (tid & 3) * 8);
            }
            ch8 qv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) qv[c] = *(const ch8*)(q + ks + c * 8);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const _Float16* row = reg + off[j];
                float a = acc[j];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const ch8 v = *(const ch8*)(row + c * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const ch2 a2 = {qv[c][2 * e], qv[c][2 * e + 1]}, b2 = {v[2 * e], v[2 * e + 1]};
                        a = __builtin_amdgcn_fdot2(a2, b2, a, false);
                    }
                }
                acc[j] = a;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) dots[pixl][part * PT + j] = acc[j];
    __syncthreads();
    if (!live) return;
    const float fx = s_fx[pixl], fy = s_fy[pixl];
    _Float16* o = out + (long)pix * ld + (long)l * N * N;
    for (int idx = part; idx < N * N; idx += 4) {
        const int a = idx / N, b2 = idx % N;                              // a: x offset, b2: y offset (the reference's meshgrid(dy, dx) order)
        const float* d0 = &dots[pixl][b2 * N1 + a];
        const float v = (1.f - fx) * (1.f - fy) * d0[0] + fx * (1.f - fy) * d0[1] + (1.f - fx) * fy * d0[N1] + fx * fy * d0[N1 + 1];
        o[idx] = (_Float16)(v * inv_sqrt_d);
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
    hipLaunchKernelGGL(k_corr_lookup<float>, dim3(cdiv(P, 4), num_levels, B), dim3(256), 0, st, fmap1, lv, coords, out, B, H, W, D, radius, cs, ps, bs,
                       1.f / sqrtf((float)D));
    TCL_LAUNCH_RET();
}

// same lookup written as f16 rows [B*H*W, ld] (channels [0, L*(2r+1)^2); the caller keeps the padding channels zero) -- the layout the
// motion encoder's first 1x1 convolution (a GEMM) reads
int tcl_corr_lookup_rows_f16(const float* fmap1, const float* const* fmap2_levels, const int* level_h, const int* level_w, int num_levels,
                             const float* coords, void* out_rows, int ld, int B, int H, int W, int D, int radius, hipStream_t st) {
    TCL_CHECK_ARG(fmap1 && fmap2_levels && level_h && level_w && coords && out_rows && B > 0 && H > 0 && W > 0);
    TCL_CHECK_ARG(num_levels >= 1 && num_levels <= 4 && radius >= 1 && radius <= 5 && D % 64 == 0 && D <= 512);
    const int n = 2 * radius + 1, P = H * W, C = num_levels * n * n;
    TCL_CHECK_ARG(ld >= C);
    CorrLevels lv;
    for (int i = 0; i < 4; ++i) { lv.f2[i] = i < num_levels ? fmap2_levels[i] : nullptr; lv.H[i] = i < num_levels ? level_h[i] : 0; lv.W[i] = i < num_levels ? level_w[i] : 0; }
    for (int i = 0; i < num_levels; ++i) TCL_CHECK_ARG(lv.f2[i] && lv.H[i] > 0 && lv.W[i] > 0);
    hipLaunchKernelGGL(k_corr_lookup<_Float16>, dim3(cdiv(P, 4), num_levels, B), dim3(256), 0, st, fmap1, lv, coords, (_Float16*)out_rows, B, H, W, D, radius,
                       (long)1, (long)ld, (long)P * ld, 1.f / sqrtf((float)D));
    TCL_LAUNCH_RET();
}

// round 6: the same rows from the tile-sharing kernel (f16 feature maps; one entry, radius 4, D % 32 == 0), then k_corr_lookup over the pixels of the tiles that
// kernel flagged (box > CT_RMAX points).  tile_flags: num_levels * ceil(H/8) * ceil(W/8) ints of scratch (written by every call).
int tcl_corr_lookup_rows_tiled_f16(const void* fmap1_h, const void* const* fmap2_levels_h, const float* fmap1, const float* const* fmap2_levels,
                                   const int* level_h, const int* level_w, int num_levels, const float* coords, void* out_rows, int ld, int H, int W, int D,
                                   int radius, int* tile_flags, hipStream_t st) {
    TCL_CHECK_ARG(fmap1_h && fmap2_levels_h && fmap1 && fmap2_levels && level_h && level_w && coords && out_rows && tile_flags && H > 0 && W > 0);
    TCL_CHECK_ARG(num_levels >= 1 && num_levels <= 4 && radius == 4 && D % 64 == 0 && D <= 512);
    const int n = 2 * radius + 1, P = H * W, C = num_levels * n * n;
    TCL_CHECK_ARG(ld >= C);
    CorrLevels lv; CorrLevelsH lh;
    for (int i = 0; i < 4; ++i) {
        lv.f2[i] = i < num_levels ? fmap2_levels[i] : nullptr; lh.f2[i] = i < num_levels ? (const _Float16*)fmap2_levels_h[i] : nullptr;
        lv.H[i] = lh.H[i] = i < num_levels ? level_h[i] : 0; lv.W[i] = lh.W[i] = i < num_levels ? level_w[i] : 0;
    }
    for (int i = 0; i < num_levels; ++i) TCL_CHECK_ARG(lv.f2[i] && lh.f2[i] && lv.H[i] > 0 && lv.W[i] > 0);
    const int tiles = ((H + 7) / 8) * ((W + 7) / 8);
    const float isd = 1.f / sqrtf((float)D);
    hipLaunchKernelGGL(k_corr_lookup_tile, dim3(tiles, num_levels), dim3(256), 0, st, (const _Float16*)fmap1_h, lh, coords, (_Float16*)out_rows, H, W, D, (long)ld, isd,
                       tile_flags);
    hipLaunchKernelGGL(k_corr_lookup<_Float16>, dim3(cdiv(P, 4), num_levels, 1), dim3(256), 0, st, fmap1, lv, coords, (_Float16*)out_rows, 1, H, W, D, radius,
                       (long)1, (long)ld, (long)P * ld, isd, (const int*)tile_flags);
    TCL_LAUNCH_RET();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// MemFlowNet encoders (core/Networks/MemFlowNet/cnn.py:124-216 BasicEncoder): the pieces the f16 NHWC GEMM / conv3x3 kernels do not
// cover -- 7x7 stride-2 stem on the 3-channel image, InstanceNorm2d, add+ReLU, stride-2 pixel subsampling for the 1x1 shortcut.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// y[b][oy][ox][0..63] (NHWC f16) = act(conv7x7(stride 2, pad 3)(x NCHW f32 [B,3,H,W]) + bias); w_t [147][64] f32, index (c*49 + ky*7 + kx)
__global__ __launch_bounds__(256) void k_conv7x7s2(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                   _Float16* __restrict__ y, int H, int W, int Ho, int Wo, int relu) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x, P = Ho * Wo;
    if (p >= P) return;
    const int oy = p / Wo, ox = p - oy * Wo;
    float acc[64];
#pragma unroll
    for (int o = 0; o < 64; ++o) acc[o] = bias[o];
    for (int c = 0; c < 3; ++c) {
        const float* src = x + ((long)b * 3 + c) * H * W;
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = oy * 2 - 3 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = ox * 2 - 3 + kx;
                if (ix < 0 || ix >= W) continue;
                const float v = src[(long)iy * W + ix];
                const float* wr = wt + (c * 49 + ky * 7 + kx) * 64;
#pragma unroll
                for (int o = 0; o < 64; ++o) acc[o] += v * wr[o];
            }
        }
    }
    _Float16* dst = y + ((long)b * P + p) * 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) { float a = acc[q * 8 + j]; v[j] = (_Float16)(relu ? fmaxf(a, 0.f) : a); }
        *(h8*)(dst + q * 8) = v;
    }
}

// InstanceNorm2d (affine=False, biased variance) statistics, deterministic: per-block partial (sum, sumsq) per channel
__global__ __launch_bounds__(256) void k_in_stats(const _Float16* __restrict__ x, int HW, int C, int rows_per_block, float* __restrict__ part) {
    __shared__ float ps[256][16];
    const int b = blockIdx.y, nchunk = C / 8;                           // C <= 256 -> nchunk <= 32
    const int tc = threadIdx.x % nchunk, tr = threadIdx.x / nchunk, rp = 256 / nchunk;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, HW);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    if (tr < rp)
        for (int row = r0 + tr; row < r1; row += rp) {
            const h8 v = *(const h8*)(x + ((long)b * HW + row) * C + tc * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s[j] += f; q[j] += f * f; }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ps[threadIdx.x][j] = s[j]; ps[threadIdx.x][8 + j] = q[j]; }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += 256) {                    // output o = k*C + c (k: 0 sum, 1 sumsq): fixed-order sum over row phases
        const int k = o / C, c = o - k * C, chunk = c >> 3, j = c & 7;
        float t = 0.f;
        for (int rr = 0; rr < rp; ++rr) t += ps[rr * nchunk + chunk][k * 8 + j];
        part[(((long)b * gridDim.x + blockIdx.x) * 2 + k) * C + c] = t;
    }
}
__global__ void k_in_reduce(const float* __restrict__ part, int nblk, int C, float inv_n, float eps, float* __restrict__ stat) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f, q = 0.f;
    for (int i = 0; i < nblk; ++i) { s += part[(((long)b * nblk + i) * 2 + 0) * C + c]; q += part[(((long)b * nblk + i) * 2 + 1) * C + c]; }
    const float mean = s * inv_n, var = fmaxf(q * inv_n - mean * mean, 0.f);
    stat[((long)b * C + c) * 2] = mean; stat[((long)b * C + c) * 2 + 1] = rsqrtf(var + eps);
}
__global__ void k_in_apply(const _Float16* __restrict__ x, const float* __restrict__ stat, _Float16* __restrict__ y, int HW, int C, int relu) {
    const int b = blockIdx.y, nchunk = C / 8;
    const long total = (long)HW * nchunk;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / nchunk; const int ch = (int)(i % nchunk) * 8;
        const h8 v = *(const h8*)(x + ((long)b * HW + row) * C + ch);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 st = *(const float2*)(stat + ((long)b * C + ch + j) * 2);
            float a = ((float)v[j] - st.x) * st.y;
            o[j] = (_Float16)(relu ? fmaxf(a, 0.f) : a);
        }
        *(h8*)(y + ((long)b * HW + row) * C + ch) = o;
    }
}
__global__ void k_add_act(const _Float16* __restrict__ a, const _Float16* __restrict__ b, _Float16* __restrict__ y, long n8, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const h8 u = *(const h8*)(a + i * 8), v = *(const h8*)(b + i * 8);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = (float)u[j] + (float)v[j];
            if (act == 3) t = fmaxf(t, 0.f); else if (act == 4) t = 0.5f * t * (1.f + erff(t * 0.70710678f));
            o[j] = (_Float16)t;
        }
        *(h8*)(y + i * 8) = o;
    }
}
__global__ void k_subsample2(const _Float16* __restrict__ x, _Float16* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, nchunk = C / 8;
    const long total = (long)B * Ho * Wo * nchunk;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nchunk) * 8; long p = i / nchunk; const int xo = (int)(p % Wo); p /= Wo; const int yo = (int)(p % Ho); const long b = p / Ho;
        *(h8*)(y + (((b * Ho + yo) * Wo + xo) * (long)C) + ch) = *(const h8*)(x + (((b * H + 2 * yo) * W + 2 * xo) * (long)C) + ch);
    }
}

extern "C" {

int tcl_conv7x7s2_c3_f16(const float* x, const float* w_t, const float* bias, void* y, int B, int H, int W, int relu, hipStream_t st) {
    TCL_CHECK_ARG(x && w_t && bias && y && B > 0 && H > 0 && W > 0);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(k_conv7x7s2, dim3(cdiv((long)Ho * Wo, 256), B), dim3(256), 0, st, x, w_t, bias, (_Float16*)y, H, W, Ho, Wo, relu);
    TCL_LAUNCH_RET();
}

static inline int in_blocks(int B, int HW) { int cap = 2048 / B; cap = cap < 4 ? 4 : (cap > 256 ? 256 : cap); int n = cdiv(HW, 64); return n > cap ? cap : n; }
size_t tcl_instnorm_workspace_bytes(int B, int C) { return ((size_t)B * 256 * 2 * C + (size_t)B * C * 2) * 4 + 256; }
int tcl_instnorm_f16(const void* x, void* y, int B, int HW, int C, float eps, int relu, void* ws, hipStream_t st) {
    TCL_CHECK_ARG(x && y && ws && B > 0 && HW > 0 && C % 8 == 0 && C <= 256 && 256 % (C / 8) == 0);
    int blocks = in_blocks(B, HW);
    const int rpb = cdiv(HW, blocks); blocks = cdiv(HW, rpb);
    float* part = (float*)ws; float* stat = part + (size_t)B * 256 * 2 * C;
    hipLaunchKernelGGL(k_in_stats, dim3(blocks, B), dim3(256), 0, st, (const _Float16*)x, HW, C, rpb, part);
    hipLaunchKernelGGL(k_in_reduce, dim3(cdiv(C, 64), B), dim3(64), 0, st, part, blocks, C, 1.f / (float)HW, eps, stat);
    const long chunks = (long)HW * (C / 8);
    hipLaunchKernelGGL(k_in_apply, dim3(stream_grid(chunks, 256, 2) > 2048 ? 2048 : stream_grid(chunks, 256, 2), B), dim3(256), 0, st,
                       (const _Float16*)x, stat, (_Float16*)y, HW, C, relu);
    TCL_LAUNCH_RET();
}
int tcl_add_act_f16(const void* a, const void* b, void* y, long n, int act, hipStream_t st) {
    TCL_CHECK_ARG(a && b && y && n > 0 && n % 8 == 0 && (act == 0 || act == 3 || act == 4));
    hipLaunchKernelGGL(k_add_act, dim3(stream_grid(n / 8, 256, 2)), dim3(256), 0, st, (const _Float16*)a, (const _Float16*)b, (_Float16*)y, n / 8, act);
    TCL_LAUNCH_RET();
}
int tcl_subsample2_nhwc_f16(const void* x, void* y, int B, int H, int W, int C, hipStream_t st) {
    TCL_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C % 8 == 0);
    hipLaunchKernelGGL(k_subsample2, dim3(stream_grid((long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8), 256, 1)), dim3(256), 0, st, (const _Float16*)x, (_Float16*)y, B, H, W, C);
    TCL_LAUNCH_RET();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// MemFlowNet update block (core/Networks/MemFlowNet/sk2.py): depthwise large-kernel convolution of PCBlock4_Deep_nopool_res fused with
// its residual and GELU, and the small f32 <-> padded-f16 glue around the GEMMs.

// y = gelu(x + depthwise_kxk(x) + bias)  (sk2.py:26-27: `x = F.gelu(x + conv(x))`); x, y [B,H,W,C] f16 NHWC, w [k*k][C] f16, k odd.
// Block = 16x16 output pixels x 8 channels: the (16+k-1)^2 input halo of those 8 channels is staged in LDS once (16 B per pixel), every
// thread then reads its k*k taps from LDS; the 8-channel weight vectors are block-uniform (scalar loads).
// acc += f16(lo | hi half of v2) * f16(lo | hi half of w2), one instruction: v_fma_mix_f32 converts the halves exactly and rounds once -- the same value as
// cvt, cvt, fma (an f16 x f16 product is exact in f32), at a third of the issue slots.  hipcc does not select it here (it emits 256 v_cvt + 96 fma / pk_fma per
// kernel row of 15 taps, round 6 ISA reading): the 15x15 depthwise convolutions were 18 % of a MemFlowNet frame pair.
__device__ __forceinline__ void fma_mix_lo(float& acc, unsigned v2, unsigned w2) { asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(v2), "s"(w2)); }
__device__ __forceinline__ void fma_mix_hi(float& acc, unsigned v2, unsigned w2) { asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(v2), "s"(w2)); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ __launch_bounds__(256) void k_dwconv_gelu(const _Float16* __restrict__ x, const _Float16* __restrict__ w, const _Float16* __restrict__ bias,
                                                     _Float16* __restrict__ y, int H, int W, int C) {
    constexpr int R = K / 2, TS = 16 + K - 1;
    __shared__ h8 tile[TS * TS];
    const int b = blockIdx.z, ch = blockIdx.y * 8, tilesx = (W + 15) / 16;
    const int ty0 = (blockIdx.x / tilesx) * 16, tx0 = (blockIdx.x % tilesx) * 16;
    const _Float16* xb = x + (long)b * H * W * C + ch;
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int iy = ty0 - R + i / TS, ix = tx0 - R + i % TS;
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const h8*)(xb + ((long)iy * W + ix) * C);
        tile[i] = v;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15, oy = ty0 + ly, ox = tx0 + lx;
    if (oy >= H || ox >= W) return;
    float acc[8];
    const h8 bv = *(const h8*)(bias + ch);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (float)bv[j];
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const u32x4 v = *(const u32x4*)&tile[(ly + ky) * TS + lx + kx];
            const u32x4 wl = *(const u32x4*)(w + (long)(ky * K + kx) * C + ch);   // block-uniform: scalar loads
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ws_ = __builtin_amdgcn_readfirstlane(wl[j]);
                fma_mix_lo(acc[2 * j], v[j], ws_);
                fma_mix_hi(acc[2 * j + 1], v[j], ws_);
            }
        }
    const h8 xc = tile[(ly + R) * TS + lx + R];
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float t = (float)xc[j] + acc[j]; o[j] = (_Float16)(0.5f * t * (1.f + erff(t * 0.70710678f))); }
    *(h8*)(y + (((long)b * H + oy) * W + ox) * C + ch) = o;
}
// k == 1: per-channel scale and bias
__global__ void k_dw1_gelu(const _Float16* __restrict__ x, const _Float16* __restrict__ w, const _Float16* __restrict__ bias, _Float16* __restrict__ y,
                           long rows, int C) {
    const int nchunk = C / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * nchunk; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nchunk) * 8;
        const h8 v = *(const h8*)(x + (i / nchunk) * C + ch), wv = *(const h8*)(w + ch), bv = *(const h8*)(bias + ch);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = (float)v[j] + (float)v[j] * (float)wv[j] + (float)bv[j]; o[j] = (_Float16)(0.5f * t * (1.f + erff(t * 0.70710678f))); }
        *(h8*)(y + (i / nchunk) * C + ch) = o;
    }
}

// f32 NCHW [B,Cs,H,W] -> f16 NHWC rows [B*H*W, ld] channels [c0, c0+Cs) (other channels untouched unless zero_rest)
__global__ void k_nchw_f32_to_rows_f16(const float* __restrict__ x, _Float16* __restrict__ y, int B, int Cs, int P, int ld, int c0, int zero_rest) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / P; const int p = (int)(i % P);
        _Float16* row = y + i * ld;
        if (zero_rest) for (int c = 0; c < ld; ++c) row[c] = (_Float16)0.f;
        for (int c = 0; c < Cs; ++c) row[c0 + c] = (_Float16)x[(b * Cs + c) * P + p];
    }
}
// f16 NHWC rows [B*P, ld] channels [c0, c0+Cs) -> f32 NCHW, out = alpha*out + beta*value
__global__ void k_rows_f16_to_nchw_f32(const _Float16* __restrict__ x, float* __restrict__ y, int B, int Cs, int P, int ld, int c0, float alpha, float beta) {
    const long total = (long)B * Cs * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % P); long t = i / P; const int c = (int)(t % Cs); const long b = t / Cs;
        const float v = (float)x[(b * P + p) * ld + c0 + c];
        y[i] = (alpha != 0.f ? alpha * y[i] : 0.f) + beta * v;
    }
}
// y = a + s*b (f16)
__global__ void k_axpy_f16(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float s, _Float16* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const h8 u = *(const h8*)(a + i * 8), v = *(const h8*)(b + i * 8);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)((float)u[j] + s * (float)v[j]);
        *(h8*)(y + i * 8) = o;
    }
}
// MemFlowNet.upsample_flow (MemFlow.py:172-183): convex combination of the 3x3 neighbourhood of 8*flow with softmax(mask) weights.
// flow [B,2,h,w] f32, mask rows [B*h*w, ldm] f16 with channel (k*64 + i*8 + j) (k: 3x3 tap, (i,j): sub-pixel), mask_scale 0.25 -> up [B,2,8h,8w]
__global__ void k_upsample_flow(const float* __restrict__ flow, const _Float16* __restrict__ mask, int ldm, float mask_scale, float* __restrict__ up,
                                int B, int h, int w) {
    const long total = (long)B * h * w * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int sub = (int)(i & 63); long p = i >> 6; const int x = (int)(p % w); p /= w; const int y = (int)(p % h); const long b = p / h;
        const _Float16* m = mask + ((b * h + y) * w + x) * (long)ldm + sub;
        float e[9], mx = -1e30f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { e[k] = mask_scale * (float)m[k * 64]; mx = fmaxf(mx, e[k]); }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { e[k] = __expf(e[k] - mx); s += e[k]; }
        float fx = 0.f, fy = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;         // F.unfold zero padding
            fx += e[k] * flow[((b * 2 + 0) * h + yy) * w + xx];
            fy += e[k] * flow[((b * 2 + 1) * h + yy) * w + xx];
        }
        const int si = sub >> 3, sj = sub & 7;
        const long o = ((long)(y * 8 + si)) * (w * 8) + x * 8 + sj;
        up[(b * 2 + 0) * (long)(h * 8) * (w * 8) + o] = 8.f * fx / s;
        up[(b * 2 + 1) * (long)(h * 8) * (w * 8) + o] = 8.f * fy / s;
    }
}

extern "C" {

int tcl_dwconv_gelu_f16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int C, int k, hipStream_t st) {
    TCL_CHECK_ARG(x && w && bias && y && B > 0 && H > 0 && W > 0 && C % 8 == 0 && (k == 1 || k == 7 || k == 15));
    const _Float16 *xp = (const _Float16*)x, *wp = (const _Float16*)w, *bp = (const _Float16*)bias;
    const dim3 grid(cdiv(H, 16) * cdiv(W, 16), C / 8, B);
    if (k == 1) hipLaunchKernelGGL(k_dw1_gelu, dim3(stream_grid((long)B * H * W * (C / 8), 256, 2)), dim3(256), 0, st, xp, wp, bp, (_Float16*)y, (long)B * H * W, C);
    else if (k == 7) hipLaunchKernelGGL(k_dwconv_gelu<7>, grid, dim3(256), 0, st, xp, wp, bp, (_Float16*)y, H, W, C);
    else hipLaunchKernelGGL(k_dwconv_gelu<15>, grid, dim3(256), 0, st, xp, wp, bp, (_Float16*)y, H, W, C);
    TCL_LAUNCH_RET();
}
int tcl_nchw_f32_to_rows_f16(const float* x, void* y, int B, int Cs, int P, int ld, int c0, int zero_rest, hipStream_t st) {
    TCL_CHECK_ARG(x && y && B > 0 && Cs > 0 && P > 0 && c0 >= 0 && c0 + Cs <= ld);
    hipLaunchKernelGGL(k_nchw_f32_to_rows_f16, dim3(stream_grid((long)B * P, 256, 1)), dim3(256), 0, st, x, (_Float16*)y, B, Cs, P, ld, c0, zero_rest);
    TCL_LAUNCH_RET();
}
int tcl_rows_f16_to_nchw_f32(const void* x, float* y, int B, int Cs, int P, int ld, int c0, float alpha, float beta, hipStream_t st) {
    TCL_CHECK_ARG(x && y && B > 0 && Cs > 0 && P > 0 && c0 >= 0 && c0 + Cs <= ld);
    hipLaunchKernelGGL(k_rows_f16_to_nchw_f32, dim3(stream_grid((long)B * Cs * P, 256, 1)), dim3(256), 0, st, (const _Float16*)x, y, B, Cs, P, ld, c0, alpha, beta);
    TCL_LAUNCH_RET();
}
int tcl_axpy_f16(const void* a, const void* b, float s, void* y, long n, hipStream_t st) {
    TCL_CHECK_ARG(a && b && y && n > 0 && n % 8 == 0);
    hipLaunchKernelGGL(k_axpy_f16, dim3(stream_grid(n / 8, 256, 2)), dim3(256), 0, st, (const _Float16*)a, (const _Float16*)b, s, (_Float16*)y, n / 8);
    TCL_LAUNCH_RET();
}
int tcl_upsample_flow_f32(const float* flow, const void* mask, int ldm, float mask_scale, float* up, int B, int h, int w, hipStream_t st) {
    TCL_CHECK_ARG(flow && mask && up && B > 0 && h > 0 && w > 0 && ldm >= 576);
    hipLaunchKernelGGL(k_upsample_flow, dim3(stream_grid((long)B * h * w * 64, 256, 1)), dim3(256), 0, st, flow, (const _Float16*)mask, ldm, mask_scale, up, B, h, w);
    TCL_LAUNCH_RET();
}

}  // extern "C"

// MemFlowNet.encode_context (MemFlow.py:112-115): c [P,256] -> net = tanh(c[:, :128]), inp = relu(c[:, 128:])
__global__ void k_context_split(const _Float16* __restrict__ c, _Float16* __restrict__ net, _Float16* __restrict__ inp, long P) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P * 16; i += (long)gridDim.x * blockDim.x) {
        const long p = i >> 4; const int ch = (int)(i & 15) * 8;
        const h8 a = *(const h8*)(c + p * 256 + ch), b = *(const h8*)(c + p * 256 + 128 + ch);
        h8 o1, o2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o1[j] = (_Float16)tanhf((float)a[j]); o2[j] = (_Float16)fmaxf((float)b[j], 0.f); }
        *(h8*)(net + p * 128 + ch) = o1; *(h8*)(inp + p * 128 + ch) = o2;
    }
}
extern "C" int tcl_context_split_f16(const void* c, void* net, void* inp, long P, hipStream_t st) {
    TCL_CHECK_ARG(c && net && inp && P > 0);
    hipLaunchKernelGGL(k_context_split, dim3(stream_grid(P * 16, 256, 1)), dim3(256), 0, st, (const _Float16*)c, (_Float16*)net, (_Float16*)inp, P);
    TCL_LAUNCH_RET();
}
