// Path 2 of TC-Light on gfx950: the two-stage temporal-consistency optimiser.
// HBM-bound f32 kernels, forward and hand-derived backward fused per loss term, plus the
// whole-stage drivers that enqueue every iteration on one HIP stream with no host sync
// (the reference synchronises on loss.item() every iteration, generate.py:431,513).
//
// Layout: images planar [n,3,H,W] f32 (the reference's NCHW), flows [N,2,H,W], masks [N,1,H,W],
// unq_inv int32 [N*H*W], codebook channel-planar [3,K] (the reference's features_dc [K,3] transposed: the gather and the
// scatter-add atomics of 64 neighbouring pixels then touch contiguous addresses per channel), exposure [N,3,4].
// Reference functions restated (file:line under /root/reference):
//   warp_flow utils/flow_utils.py:5-16 · relaxed_ms_ssim utils/loss_utils.py:73-211 · TVLoss :324-340
//   l1_loss :25-26 · exposure_align generate.py:354-451 · unique_tensor_optimization :453-533
//   OptDataset.exposure_align utils/dataloader.py:38-42 · SH2RGB/RGB2SH utils/sh_utils.py:114-117
#include "common.h"
#include "../../include/tclight_hip.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

#define SH_C0 0.28209479177387814f
#define ACC_SLOTS 32   // loss accumulators: [ACC_SLOTS][4] fixed-point cells = {l1, tv_h, tv_w, flow}

// ---- run-to-run determinism (round 3).  Every reduction whose order the hardware chooses -- block partial sums meeting in one cell, the
// bicubic scatter of the warp's backward pass -- accumulates in 64-bit FIXED POINT: integer addition is associative, so the result does not
// depend on the order the atomics land in, and two runs (or two ranks replicating the optimiser) produce the same bits.  The scale of each
// accumulator is a power of two chosen from the quantity's range (value * 2^k is exact in f32; the conversion rounds to 2^-k once per
// contribution, far below the f32 rounding of the float sums it replaces).  Codebook rows are accumulated without atomics at all: the
// track ids of get_flowid are unique within a frame (tcl_track_ids_unique verifies it once per run), so one frame of the mini-batch at a time
// is a conflict-free read-modify-write, and the frames go in a fixed order.
typedef long long fx_t;
#define FX_SSIM 1073741824.f            // 2^30: per-block sums of SSIM values, |.| <= TW*TH
#define FX_ACC 1048576.f                // 2^20: loss sums over a mini-batch (<= ~1e8)
#define FX_FLOW_SHIFT 22                // mask * bicubic weights landing on one pixel of the previous frame, in a 32-BIT cell (round 4; 2^32 in 64-bit
                                        // cells before): at 2^22 per unit the resolution is 2.4e-7 per contribution and the range +-512 -- a cell receives
                                        // ~16 weights of |w| <= 1 from a smooth flow; 32-bit integer atomics run ~2x the rate of 64-bit ones and gpre
                                        // halves (memset, reads).  Round 5: the scale is PER FRAME, 2^flow_shift[f] with flow_shift[f] =
                                        // min(22, 30 - ceil(log2 L_f)), L_f = the largest number of masked-in pixels of frame f whose 4x4 tap window
                                        // covers one cell (tcl_flow_cell_shift, once per clip: flows and masks do not change during the optimisation).
                                        // |mask * weight| <= 1 per pixel and cell, so no cell -- LDS window or global -- can leave the 32-bit range
                                        // however strongly a flow converges (a zoom-out, a degenerate flow net output): frames past 512 pixels per cell
                                        // trade resolution for range instead of wrapping silently.  Smooth flows get 22 everywhere: round 4's bits.
typedef int fxq_t;
#define FX_EXPO 281474976710656.f       // 2^48: exposure gradient components (sums of image * pixel gradient)
__device__ __forceinline__ void fx_add(fx_t* p, float v, float scale) {
    atomicAdd((unsigned long long*)p, (unsigned long long)__float2ll_rn(v * scale));
}
__device__ __forceinline__ float fx_get(fx_t v, float inv_scale) { return (float)((double)v * (double)inv_scale); }
#define TW 32
#define TH 16
#define HALO 10
#define TIW (TW + HALO)
#define TIH (TH + HALO)

struct Gauss11 { float g[11]; };
static Gauss11 make_gauss() {
    Gauss11 G; float s = 0.f;
    for (int i = 0; i < 11; ++i) { float c = (float)(i - 5); G.g[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f)); s += G.g[i]; }
    for (int i = 0; i < 11; ++i) G.g[i] /= s;
    return G;
}

#include "bicubic.h"
#include <type_traits>
// KEEP(x) pins a loaded value at its place in the program: hipcc otherwise sinks a load whose only use sits behind a guard into that guard
// (a branch and a s_waitcnt of its own per load -- the loads of a thread then run one round trip after the other).
#define KEEP(x) asm volatile("" :: "v"(x))

__global__ void k_warp_fwd(const float* __restrict__ img, const float* __restrict__ flow, float* __restrict__ out,
                           int C, int H, int W, int flow_c) {
    const int P = H * W, n = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        int y = p / W, x = p - y * W;
        const float* fl = flow + (size_t)n * flow_c * P;
        Tap t = make_tap(fl[p], fl[P + p], x, y, W, H);
        for (int c = 0; c < C; ++c) {
            const float* pl = img + ((size_t)n * C + c) * P;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int yy = t.y0 + j; if (yy < 0 || yy >= H) continue;
                float r = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { int xx = t.x0 + i; if (xx >= 0 && xx < W) r += t.wx[i] * pl[yy * W + xx]; }
                acc += t.wy[j] * r;
            }
            out[((size_t)n * C + c) * P + p] = acc;
        }
    }
}
__global__ void k_warp_bwd(const float* __restrict__ gout, const float* __restrict__ flow, float* __restrict__ gimg,
                           int C, int H, int W, int flow_c) {
    const int P = H * W, n = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        int y = p / W, x = p - y * W;
        const float* fl = flow + (size_t)n * flow_c * P;
        Tap t = make_tap(fl[p], fl[P + p], x, y, W, H);
        for (int c = 0; c < C; ++c) {
            float g = gout[((size_t)n * C + c) * P + p];
            float* pl = gimg + ((size_t)n * C + c) * P;
            for (int j = 0; j < 4; ++j) {
                int yy = t.y0 + j; if (yy < 0 || yy >= H) continue;
                for (int i = 0; i < 4; ++i) { int xx = t.x0 + i; if (xx >= 0 && xx < W) atomicAdd(pl + yy * W + xx, t.wy[j] * t.wx[i] * g); }
            }
        }
    }
}

// ---------------------------------------------------------------- parameter -> image
// out[j] = clamp(src[idx[j]] @ M[:3,:3] + M[:3,3]) (generate.py:405-407)
__global__ void k_apply_exposure(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ expo,
                                 float* __restrict__ out, int P) {
    const int j = blockIdx.y, f = idx ? idx[j] : j;
    const float* M = expo + (size_t)f * 12;
    float m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = M[i];
    const float* s = src + (size_t)f * 3 * P; float* o = out + (size_t)j * 3 * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        float x0 = s[p], x1 = s[P + p], x2 = s[2 * P + p];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = x0 * m[c] + x1 * m[4 + c] + x2 * m[8 + c] + m[c * 4 + 3];
            o[c * P + p] = fminf(fmaxf(t, 0.f), 1.f);
        }
    }
}
// d(loss)/dM of cat row j from the gradient of its clamped output -> efx[j][12] (fixed point; ordered add into grad_expo by k_expo_fin).
// Rows j >= b are the "previous frame" images: their gradient is the flow term's scatter, held in fixed point (gpre, see k_flow_loss).
__global__ void k_exposure_bwd(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ expo,
                               const float* __restrict__ gimg, const fxq_t* __restrict__ gpre, float pre_scale, const int* __restrict__ flow_shift,
                               int b, fx_t* __restrict__ efx, int P) {
    __shared__ float red[16];
    const int j = blockIdx.y, f = idx[j];
    if (j >= b) pre_scale = __builtin_ldexpf(pre_scale, -flow_shift[idx[j - b]]);      // the scatter of slot j - b ran at its current frame's scale
    const float* M = expo + (size_t)f * 12;
    float m[12], acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { m[i] = M[i]; acc[i] = 0.f; }
    const float* s = src + (size_t)f * 3 * P;
    const float* g = j < b ? gimg + (size_t)j * 3 * P : nullptr;
    const fxq_t* gq = j < b ? nullptr : gpre + (size_t)(j - b) * 3 * P;
    // (image or pre-image gradient: decided once per block, outside the pixel loop -- a per-load select on a run-time condition is a branch and a wait per load)
    auto body = [&](auto PRE) {
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
            float x[3] = {s[p], s[P + p], s[2 * P + p]}, gv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gv[c] = decltype(PRE)::value ? (float)gq[c * P + p] * pre_scale : g[c * P + p];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = x[0] * m[c] + x[1] * m[4 + c] + x[2] * m[8 + c] + m[c * 4 + 3];
                float gc = (t >= 0.f && t <= 1.f) ? gv[c] : 0.f;
                acc[c] += x[0] * gc; acc[4 + c] += x[1] * gc; acc[8 + c] += x[2] * gc; acc[c * 4 + 3] += gc;
            }
        }
    };
    if (j < b) body(std::false_type{}); else body(std::true_type{});
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float r = block_sum(acc[i], red);
        if (threadIdx.x == 0 && r != 0.f) fx_add(efx + (size_t)j * 12 + i, r, FX_EXPO);
    }
}
// grad_expo[idx[j]] += efx[j], rows in order (a frame may sit in the batch twice: as a current and as a previous frame)
__global__ void k_expo_fin(const fx_t* __restrict__ efx, const int* __restrict__ idx, int rows, float* __restrict__ gexpo) {
    const int i = threadIdx.x;
    if (i >= 12) return;
    for (int j = 0; j < rows; ++j) gexpo[(size_t)idx[j] * 12 + i] += fx_get(efx[(size_t)j * 12 + i], 1.f / FX_EXPO);
}
// Memory-level parallelism for k_codebook_bwd (round 5, second pass).  Its first form walked a thread's pixels one after the other, each a chain of
// dependent round trips (index -> gradient / mask -> read-modify-write of the row), with the block count already at the occupancy limit.  Now a thread
// takes P2_U pixels per trip and walks the chain ONCE for all of them; KEEP(x) pins a loaded value at its place in the program: hipcc otherwise sinks a
// load whose only use sits behind a per-pixel guard into that guard, which serialises the round trips again.  18.7 -> 12.5 us per launch (x 32 per
// iteration), same bits.  The same rewrite of the two lazy-Adam kernels (k_adam_catchup_frame / k_adam_touched_frame: index -> step counter -> atomic
// claim -> 9-12 row loads) was measured and REVERTED: at K = 2.7e8 rows they get slower the more rows a thread has in flight -- 846 / 857 us per launch
// at 1 pixel per trip, 868 / 940 at 2, 1060 / 997 at 4, 1916 / 1527 at 8 (profiles/r5b_prof_p2_adjacent.txt; pixels a grid stride or a block width
// apart alike, profiles/r5b_prof_p2_stride.txt) -- they are bound by the scattered 4-byte traffic itself (atomics and 9-12 planes per row), not by latency.
#ifndef P2_U
#define P2_U 4
#endif
// out[j] = clamp(SH2RGB(feat[inv[fidx[j]*P + p]])) (generate.py:499-501)
// cmask (may be null): bit c of byte [j][p] = channel c of that pixel lies inside [0, 1] -- the clamp's gradient mask, handed to k_codebook_bwd so that it
// need not gather the codebook row a second time (round 5: a third of that kernel's traffic).
__global__ void k_gather_codebook(const float* __restrict__ feat, const int* __restrict__ inv, const int* __restrict__ fidx,
                                  float* __restrict__ out, int P, size_t K, unsigned char* __restrict__ cmask) {
    const int j = blockIdx.y, f = fidx ? fidx[j] : j;
    const int* iv = inv + (size_t)f * P; float* o = out + (size_t)j * 3 * P;
    // 4 pixels per trip, a block width apart: 4 index loads, then the 12 row loads, in flight together (round 5, second pass; the first form walked
    // index -> row pixel by pixel)
    const int stride = gridDim.x * blockDim.x;
    for (int p0 = blockIdx.x * (blockDim.x * 4) + threadIdx.x; p0 < P; p0 += 4 * stride) {
        size_t id[4]; float fv[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = (size_t)iv[min(p0 + u * (int)blockDim.x, P - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) fv[u][c] = feat[c * K + id[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) KEEP(fv[u][c]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * (int)blockDim.x;
            if (p >= P) break;
            int mk = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float val = fv[u][c] * SH_C0 + 0.5f;
                mk |= (val >= 0.f && val <= 1.f) ? (1 << c) : 0;
                o[c * P + p] = fminf(fmaxf(val, 0.f), 1.f);
            }
            if (cmask) cmask[(size_t)j * P + p] = (unsigned char)mk;
        }
    }
}
// d(loss)/d(codebook): cat row j0 + blockIdx.y.  ATOMIC == false: the ids of one frame are distinct, so a launch over ONE row is a conflict-free
// read-modify-write and the rows of a mini-batch are launched one after the other (fixed order: deterministic, and no atomic unit in the way);
// ATOMIC == true (ids that repeat inside a frame): all rows in one launch, float atomics, order not reproducible.
// cmask != null (lazy schedule, round 5): the clamp test of the gathered value comes from the mask byte the lazy gather wrote (bit c: channel c
// inside [0, 1]) -- the codebook row in memory may still be steps behind, and the mask saves this kernel's three feat reads per pixel.
// The run-time-uniform choices (image or pre-image gradient, clamp mask or codebook read) are template parameters of the body: a per-load select on a
// run-time condition, even a wave-uniform one, makes hipcc branch around each load and wait for it (cdna_hip_programming.md, trap (c)).
template <bool ATOMIC, bool PRE, bool HASCM>
__device__ __forceinline__ void codebook_bwd_body(const float* __restrict__ feat, const int* __restrict__ iv, const float* __restrict__ g,
                                                  const fxq_t* __restrict__ gq, float pre_scale, float* __restrict__ gfeat, int P, size_t K,
                                                  const unsigned char* __restrict__ cm) {
    const int stride = gridDim.x * blockDim.x;
    for (int p0 = blockIdx.x * (blockDim.x * P2_U) + threadIdx.x; p0 < P; p0 += P2_U * stride) {
        int px[P2_U]; bool in[P2_U]; size_t id[P2_U]; int mk[P2_U]; float gc[P2_U][3];
#pragma unroll
        for (int u = 0; u < P2_U; ++u) { in[u] = p0 + u * (int)blockDim.x < P; px[u] = in[u] ? p0 + u * (int)blockDim.x : p0; }
#pragma unroll
        for (int u = 0; u < P2_U; ++u) {
            id[u] = (size_t)iv[px[u]];
            mk[u] = HASCM ? cm[px[u]] : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) gc[u][c] = PRE ? (float)gq[c * P + px[u]] * pre_scale : g[c * P + px[u]];
        }
        float fv[P2_U][3], old[P2_U][3];
#pragma unroll
        for (int u = 0; u < P2_U; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (!HASCM) fv[u][c] = feat[c * K + id[u]];
                if (!ATOMIC) old[u][c] = gfeat[c * K + id[u]];      // the ids of a frame are distinct: 12 independent read-modify-writes
            }
#pragma unroll
        for (int u = 0; u < P2_U; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) { if (!HASCM) KEEP(fv[u][c]); if (!ATOMIC) KEEP(old[u][c]); }
#pragma unroll
        for (int u = 0; u < P2_U; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                bool inr;
                if (HASCM) inr = (mk[u] >> c) & 1;
                else { const float v = fv[u][c] * SH_C0 + 0.5f; inr = v >= 0.f && v <= 1.f; }
                if (in[u] && inr && gc[u][c] != 0.f) {
                    if (ATOMIC) atomicAdd(gfeat + c * K + id[u], gc[u][c] * SH_C0);
                    else gfeat[c * K + id[u]] = old[u][c] + gc[u][c] * SH_C0;
                }
            }
    }
}
template <bool ATOMIC>
__global__ void k_codebook_bwd(const float* __restrict__ feat, const int* __restrict__ inv, const int* __restrict__ fidx,
                               const float* __restrict__ gimg, const fxq_t* __restrict__ gpre, float pre_scale, const int* __restrict__ flow_shift,
                               int b, int j0, float* __restrict__ gfeat, int P, size_t K, const unsigned char* __restrict__ cmask) {
    const int j = j0 + blockIdx.y, f = fidx[j];
    const int* iv = inv + (size_t)f * P;
    const unsigned char* cm = cmask ? cmask + (size_t)j * P : nullptr;
    if (j < b) {
        const float* g = gimg + (size_t)j * 3 * P;
        if (cm) codebook_bwd_body<ATOMIC, false, true>(feat, iv, g, nullptr, 0.f, gfeat, P, K, cm);
        else codebook_bwd_body<ATOMIC, false, false>(feat, iv, g, nullptr, 0.f, gfeat, P, K, cm);
    } else {
        pre_scale = __builtin_ldexpf(pre_scale, -flow_shift[fidx[j - b]]);
        const fxq_t* gq = gpre + (size_t)(j - b) * 3 * P;
        if (cm) codebook_bwd_body<ATOMIC, true, true>(feat, iv, nullptr, gq, pre_scale, gfeat, P, K, cm);
        else codebook_bwd_body<ATOMIC, true, false>(feat, iv, nullptr, gq, pre_scale, gfeat, P, K, cm);
    }
}
// Are the ids of every frame distinct?  Per frame: every pixel writes its index into scratch[id], then checks that it is still there.
__global__ void k_ids_mark(const int* __restrict__ inv, int P, int* __restrict__ scratch) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) scratch[inv[p]] = p;
}
__global__ void k_ids_check(const int* __restrict__ inv, int P, const int* __restrict__ scratch, int* __restrict__ flag) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x)
        if (scratch[inv[p]] != p) *flag = 0;
}

// ---------------------------------------------------------------- MS-SSIM
// avg_pool2d(k=2, padding=size%2, count_include_pad) of nplanes planes; if fidx != null the source
// plane of output plane q is (fidx[q/3]*3 + q%3) (used to pool the target frames in place).
__global__ void k_pool2(const float* __restrict__ in, const int* __restrict__ fidx, float* __restrict__ out,
                        int h, int w, int oh, int ow) {
    const int q = blockIdx.y, ph = h & 1, pw = w & 1;
    const float* s = in + (size_t)(fidx ? fidx[q / 3] * 3 + q % 3 : q) * h * w;
    float* o = out + (size_t)q * oh * ow;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < oh * ow; i += gridDim.x * blockDim.x) {
        int oy = i / ow, ox = i - oy * ow, y0 = 2 * oy - ph, x0 = 2 * ox - pw;
        float a = 0.f, vv[4]; bool ok[4];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int y = y0 + dy, x = x0 + dx;
                vv[dy * 2 + dx] = s[min(max(y, 0), h - 1) * w + min(max(x, 0), w - 1)];      // unconditional load, select behind it (a + 0 = a)
                ok[dy * 2 + dx] = y >= 0 && y < h && x >= 0 && x < w;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) KEEP(vv[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) a += ok[k] ? vv[k] : 0.f;
        o[i] = a * 0.25f;
    }
}
// One SSIM level: 'valid' separable 11-tap Gaussian of X, Y, XX, YY, XY through LDS, cs (and
// luminance on the last level) map -> per-plane sum, and the three local-derivative maps
// (d/d sigma12, d/d sigma1^2, d/d mu1-equivalent) the backward pass filters back.
__global__ __launch_bounds__(256) void k_ssim_fwd(const float* __restrict__ X, const float* __restrict__ Y, int h, int w, int last,
                                                  float c1, float c2, Gauss11 G, float* __restrict__ mA, float* __restrict__ mB,
                                                  float* __restrict__ mC, fx_t* __restrict__ sums) {
    __shared__ float sx[TIH][TIW + 1], sy[TIH][TIW + 1];
    __shared__ float hz[5][TIH][TW];
    __shared__ float red[16];
    const int q = blockIdx.z, oh = h - HALO, ow = w - HALO, tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const float* xp = X + (size_t)q * h * w; const float* yp = Y + (size_t)q * h * w;
    // staging: every load unconditional at a clamped address, the select behind it, the loop unrolled -- all of a thread's loads are in flight together
    // (`in ? xp[..] : 0` was a branch and a wait per load: round 5, second pass)
#pragma unroll
    for (int i0 = 0; i0 < TIH * TIW; i0 += 256) {
        const int i = i0 + threadIdx.x;
        if (i0 + 256 > TIH * TIW && i >= TIH * TIW) break;
        int r = i / TIW, c = i - r * TIW, gy = ty0 + r, gx = tx0 + c;
        bool in = gy < h && gx < w;
        const int a = min(gy, h - 1) * w + min(gx, w - 1);
        const float vx = xp[a], vy = yp[a];
        sx[r][c] = in ? vx : 0.f;
        sy[r][c] = in ? vy : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TIH * TW; i += 256) {
        int r = i / TW, c = i - r * TW;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float x = sx[r][c + k], y = sy[r][c + k], g = G.g[k];
            a0 += g * x; a1 += g * y; a2 += g * x * x; a3 += g * y * y; a4 += g * x * y;
        }
        hz[0][r][c] = a0; hz[1][r][c] = a1; hz[2][r][c] = a2; hz[3][r][c] = a3; hz[4][r][c] = a4;
    }
    __syncthreads();
    float acc = 0.f;
    for (int o = threadIdx.x; o < TH * TW; o += 256) {
        int r = o / TW, c = o - r * TW, oy = ty0 + r, ox = tx0 + c;
        if (oy >= oh || ox >= ow) continue;
        float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float g = G.g[k];
            mu1 += g * hz[0][r + k][c]; mu2 += g * hz[1][r + k][c]; e11 += g * hz[2][r + k][c];
            e22 += g * hz[3][r + k][c]; e12 += g * hz[4][r + k][c];
        }
        float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        float An = 2.f * s12 + c2, Bd = s11 + s22 + c2, cs = An / Bd;
        float da = 2.f / Bd, db = -An / (Bd * Bd), dmu = 0.f, val = cs;
        if (last) {
            float Ln = 2.f * mu1 * mu2 + c1, Ld = mu1 * mu1 + mu2 * mu2 + c1, l = Ln / Ld;
            val = l * cs;
            dmu = cs * (2.f * mu2 * Ld - Ln * 2.f * mu1) / (Ld * Ld);
            da *= l; db *= l;
        }
        size_t mi = ((size_t)q * oh + oy) * ow + ox;
        mA[mi] = da; mB[mi] = db; mC[mi] = dmu - da * mu2 - 2.f * db * mu1;
        acc += val;
    }
    float r = block_sum(acc, red);
    if (threadIdx.x == 0) fx_add(sums + q, r, FX_SSIM);
}
// per-plane product over levels, loss value and upstream scalars (loss_utils.py:196-211).
// sums: [4][planes] (levels 1..4 are computed, level 0 is the constant 1 of start_level=1);
// cnt[l] = valid pixels of level l.  scal[l][q] = d(loss)/d(sum_l[q]); loss_out += lambda*(1-mean msssim)
struct Cnt4 { float c[4]; };
__global__ void k_msssim_finalize(const fx_t* __restrict__ sums, int planes, int planes_norm, Cnt4 cn, float lambda,
                                  float* __restrict__ scal, float* __restrict__ loss_out) {
    // planes_norm: the plane count the mean runs over (== planes on one GPU; the GLOBAL batch*3 when the mini-batch is split over ranks:
    // the per-rank terms then add up to the global loss and the gradients carry the global 1/planes).
    __shared__ float red[16];
    const float wgt[4] = {0.2856f, 0.3001f, 0.2363f, 0.1333f};
    float tot = 0.f;
    for (int q = threadIdx.x; q < planes; q += blockDim.x) {
        float v[4], prod = 1.f;
        for (int l = 0; l < 4; ++l) { v[l] = fmaxf(fx_get(sums[l * planes + q], 1.f / FX_SSIM) / cn.c[l], 0.f); prod *= powf(v[l], wgt[l]); }
        tot += prod;
        for (int l = 0; l < 4; ++l)
            scal[l * planes + q] = v[l] > 0.f ? -lambda / (float)planes_norm * prod * wgt[l] / v[l] / cn.c[l] : 0.f;
    }
    float r = block_sum(tot, red);
    if (threadIdx.x == 0) *loss_out = planes_norm == planes ? lambda * (1.f - r / (float)planes) : lambda * ((float)planes - r) / (float)planes_norm;
}
// grad wrt X of one level: s * (Y*G^T(a) + 2X*G^T(b) + G^T(c)) + avgpool-backward of the next level's grad
__global__ __launch_bounds__(256) void k_ssim_bwd(const float* __restrict__ X, const float* __restrict__ Y, int h, int w,
                                                  const float* __restrict__ mA, const float* __restrict__ mB, const float* __restrict__ mC,
                                                  const float* __restrict__ scal, Gauss11 G, const float* __restrict__ gnext,
                                                  int nh, int nw, float* __restrict__ gout) {
    __shared__ float sm[3][TIH][TIW + 1];
    __shared__ float hz[3][TIH][TW];
    const int q = blockIdx.z, oh = h - HALO, ow = w - HALO, tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const size_t mo = (size_t)q * oh * ow;
#pragma unroll
    for (int i0 = 0; i0 < TIH * TIW; i0 += 256) {          // (unconditional clamped loads + selects, unrolled: as in k_ssim_fwd)
        const int i = i0 + threadIdx.x;
        if (i0 + 256 > TIH * TIW && i >= TIH * TIW) break;
        int r = i / TIW, c = i - r * TIW, my = ty0 - HALO + r, mx = tx0 - HALO + c;
        bool in = my >= 0 && my < oh && mx >= 0 && mx < ow;
        size_t mi = mo + (size_t)min(max(my, 0), oh - 1) * ow + min(max(mx, 0), ow - 1);
        const float va = mA[mi], vb = mB[mi], vc = mC[mi];
        sm[0][r][c] = in ? va : 0.f; sm[1][r][c] = in ? vb : 0.f; sm[2][r][c] = in ? vc : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TIH * TW; i += 256) {
        int r = i / TW, c = i - r * TW;
        float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) { float g = G.g[k]; a0 += g * sm[0][r][c + k]; a1 += g * sm[1][r][c + k]; a2 += g * sm[2][r][c + k]; }
        hz[0][r][c] = a0; hz[1][r][c] = a1; hz[2][r][c] = a2;
    }
    __syncthreads();
    const float s = scal[q];
    for (int o = threadIdx.x; o < TH * TW; o += 256) {
        int r = o / TW, c = o - r * TW, y = ty0 + r, x = tx0 + c;
        if (y >= h || x >= w) continue;
        float ta = 0, tb = 0, tc = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) { float g = G.g[k]; ta += g * hz[0][r + k][c]; tb += g * hz[1][r + k][c]; tc += g * hz[2][r + k][c]; }
        size_t pi = (size_t)q * h * w + (size_t)y * w + x;
        float g = s * (Y[pi] * ta + 2.f * X[pi] * tb + tc);
        if (gnext) g += 0.25f * gnext[(size_t)q * nh * nw + (size_t)((y + (h & 1)) >> 1) * nw + ((x + (w & 1)) >> 1)];
        gout[pi] = g;
    }
}

// ---------------------------------------------------------------- per-pixel losses on `images`
// gimg[p] = avgpool-backward(grad of level 1) + L1-photometric grad + TV grad; sums -> acc[0..2]
__global__ void k_pixel_losses(const float* __restrict__ img, const float* __restrict__ tgt, const int* __restrict__ idx,
                               const float* __restrict__ g1, int h1, int w1, int H, int W, float coef_l1, float coef_tvh,
                               float coef_tvw, float* __restrict__ gimg, fx_t* __restrict__ acc) {
    __shared__ float red[16];
    const int q = blockIdx.y, P = H * W;
    const float* x = img + (size_t)q * P;
    const float* t = tgt ? tgt + ((size_t)idx[q / 3] * 3 + q % 3) * P : nullptr;
    float s_l1 = 0, s_h = 0, s_w = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        int y = p / W, xx = p - y * W;
        float v = x[p];
        // (g1 / t may be null: the load then goes to a valid dummy address of x and the select drops it -- no branch around a load)
        const float g1v = (g1 ? g1 + (size_t)q * h1 * w1 : x)[(size_t)((y + (H & 1)) >> 1) * w1 + ((xx + (W & 1)) >> 1)];
        const float tv = (t ? t : x)[p];
        KEEP(g1v); KEEP(tv);
        float g = g1 ? 0.25f * g1v : 0.f;
        if (coef_l1 != 0.f) {
            float d = v - tv;
            s_l1 += fabsf(d);
            g += coef_l1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (coef_tvh != 0.f) {
            // the four neighbours are read unconditionally (clamped to p at the border) before the first is used: a load inside `if (y > 0)` is a branch
            // and a wait of its own -- four dependent round trips per pixel in the first form (round 5, second pass); same arithmetic, same order
            const float nu = x[y > 0 ? p - W : p], nd = x[y < H - 1 ? p + W : p], nl = x[xx > 0 ? p - 1 : p], nr = x[xx < W - 1 ? p + 1 : p];
            KEEP(nu); KEEP(nd); KEEP(nl); KEEP(nr);
            if (y > 0) { float d = v - nu; s_h += d * d; g += coef_tvh * 2.f * d; }
            if (y < H - 1) g -= coef_tvh * 2.f * (nd - v);
            if (xx > 0) { float d = v - nl; s_w += d * d; g += coef_tvw * 2.f * d; }
            if (xx < W - 1) g -= coef_tvw * 2.f * (nr - v);
        }
        gimg[(size_t)q * P + p] = g;
    }
    float r0 = block_sum(s_l1, red), r1 = block_sum(s_h, red), r2 = block_sum(s_w, red);
    if (threadIdx.x == 0) {
        fx_t* a = acc + ((blockIdx.x + blockIdx.y) & (ACC_SLOTS - 1)) * 4;     // spread same-address atomics over slots
        if (coef_l1 != 0.f) fx_add(a + 0, r0, FX_ACC);
        if (coef_tvh != 0.f) { fx_add(a + 1, r1, FX_ACC); fx_add(a + 2, r2, FX_ACC); }
    }
}
// flow-consistency term (generate.py:420-427): warp(pre)*m vs img*m, fwd + bwd fused.
// images = cat[0..b), pre = cat[b..2b).  gimg [b,3,P] gets '-=' (owner pixel); the scatter into the previous frame's gradient goes to
// gpre [b,3,P] in fixed point, UNSCALED (sign * mask * bicubic weight; the consumer multiplies by scale / FX_FLOW): integer atomics.
// Round 3, second half: the scatter goes through an LDS window.  A block owns a FT_W x FT_H pixel tile; the taps of a smooth flow field land in
// the tile shifted by the flow plus the bicubic extent, so the block accumulates them in a (FT_W + 3 + slack) x (FT_H + 3 + slack) x 3
// window of integer LDS cells (64-bit in round 3, 32-bit since round 4: FX_FLOW) placed at the minimum tap origin of the tile, and flushes the non-zero cells once: ~5 global atomics per pixel
// instead of 12, every one of them to consecutive cells of a row.  Taps outside the window (flow discontinuities) go to global memory
// directly.  Integer sums: the result does not depend on which way an addend took, nor on the order -- and with W % 64 == 0 the waves
// cover the same 64-pixel row segments as the untiled form (TILED == false, kept for A/B: TCL_FLOW_TILED=0), so the two agree bit for bit.
#define FT_W 64
#ifndef FT_H
#define FT_H 16
#endif
#define FT_SL 5
#define FT_WX (FT_W + 3 + FT_SL)
#define FT_WY (FT_H + 3 + FT_SL)
struct FlowWin { int* cells; int x0, y0; float fx; };       // fx: this frame's fixed-point scale 2^flow_shift[f]
template <bool TILED>
__device__ __forceinline__ void flow_sink(const FlowWin& w, fxq_t* __restrict__ gp, int c, int P, int W, int yy, int xx, float v) {
    const int q = __float2int_rn(v * w.fx);
    if (TILED) {
        const int lx = xx - w.x0, ly = yy - w.y0;
        if ((unsigned)lx < (unsigned)FT_WX && (unsigned)ly < (unsigned)FT_WY) { atomicAdd(w.cells + (c * FT_WY + ly) * FT_WX + lx, q); return; }
    }
    atomicAdd(gp + (size_t)c * P + (size_t)yy * W + xx, q);
}
// one pixel (x, y) of cat row j: forward term, owner-pixel gradient, scatter of the pre-image gradient.  Returns sum_c |d|.
template <bool TILED>
__device__ __forceinline__ float flow_pixel(const float* __restrict__ img, const float* __restrict__ pre, const float* __restrict__ fl,
                                            const float* __restrict__ mk, float* __restrict__ gi, fxq_t* __restrict__ gp, const FlowWin& win,
                                            int x, int y, bool live, int lane, int H, int W, float scale) {
    const int P = H * W;
    const int pc = live ? y * W + x : P - 1;
    if (!live) { y = (P - 1) / W; x = P - 1 - y * W; }
    Tap t = make_tap(fl[pc], fl[P + pc], x, y, W, H);
    const float m = live ? mk[pc] : 0.f;
    float wv[3] = {0.f, 0.f, 0.f}, s = 0.f;
    // The 16 x 3 tap reads go out UNCONDITIONALLY, all before the first is used: a tap outside the image reads a clamped address with weight 0 (wv + 0 * v
    // = wv, bit for bit -- frames are finite), same tap order.  The first form skipped such taps with `continue`, which hipcc compiled to a branch and a
    // s_waitcnt vmcnt(0) per tap: 16 dependent round trips per pixel (round 5, second pass; ISA of k_flow_loss<true>).
    float pv[16][3], wt[16];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int yy = t.y0 + jj; const bool yok = yy >= 0 && yy < H; const int yc = min(max(yy, 0), H - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xx = t.x0 + i; const bool ok = yok && xx >= 0 && xx < W; const int a = yc * W + min(max(xx, 0), W - 1);
            wt[jj * 4 + i] = ok ? t.wy[jj] * t.wx[i] : 0.f;
            pv[jj * 4 + i][0] = pre[a]; pv[jj * 4 + i][1] = pre[P + a]; pv[jj * 4 + i][2] = pre[2 * P + a];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { wv[0] += wt[k] * pv[k][0]; wv[1] += wt[k] * pv[k][1]; wv[2] += wt[k] * pv[k][2]; }
    float gw[3], iv[3], go[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { iv[c] = img[c * P + pc]; go[c] = gi[c * P + pc]; }      // (with the tap reads: pc is a valid address on dead lanes too)
#pragma unroll
    for (int c = 0; c < 3; ++c) KEEP(go[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float d;
        { _Pragma("clang fp contract(off)") d = wv[c] * m - iv[c] * m; }      // (two products and a difference in every instantiation: the sign of d is the gradient)
        s += fabsf(d);
        gw[c] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m;              // 0 on dead lanes (m = 0); unscaled
        if (live) gi[c * P + pc] = go[c] - gw[c] * scale;
    }
    // Scatter of d(loss)/d(warped) into the pre-image gradient: 16 taps x 3 channels per pixel.  Neighbouring pixels of a row whose taps
    // share the integer offset (dx, dy) hit neighbouring cells: lane l's tap i and lane l+i's tap 0 are the SAME cell, so the four
    // column taps are merged across lanes with shuffles and one add per (row, channel) is issued by the lane whose tap 0 owns the
    // cell (12 adds per pixel instead of 48).  The merge is SEGMENTED by the class (y, dx, dy): a lane only absorbs taps of the
    // lanes of its own class and emits itself the taps no successor of its class absorbs -- so a wave that straddles an integer crossing
    // of a smooth flow field, a row end or the image end still takes this path (an earlier all-or-nothing version fell back to 48
    // atomics per pixel for the whole wave at every such crossing: 12x slower on flows that hover around an integer).
    const int dx = t.x0 - x, dy = t.y0 - y;
    const int ca = live ? ((dx << 16) | (dy & 0xffff)) : (int)0x80000000, cb = live ? y : -1 - lane;
    bool up[4], dn[4];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        // (all four shuffles unconditionally, combined with '&': a short-circuit '&&' would run them under a partial exec mask)
        const int ua = __shfl_up(ca, i, 64), ub = __shfl_up(cb, i, 64), da = __shfl_down(ca, i, 64), db = __shfl_down(cb, i, 64);
        up[i] = (lane >= i) & (ua == ca) & (ub == cb);          // lane l-i is of my class: I absorb its tap i
        dn[i] = (lane + i <= 63) & (da == ca) & (db == cb);     // lane l+i is of my class: it absorbs my tap i
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int yy = t.y0 + jj;
        const bool rowok = live && yy >= 0 && yy < H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gwc = t.wy[jj] * gw[c];
            const float v0 = gwc * t.wx[0], v1 = gwc * t.wx[1], v2 = gwc * t.wx[2], v3 = gwc * t.wx[3];
            const float u1 = __shfl_up(v1, 1, 64), u2 = __shfl_up(v2, 2, 64), u3 = __shfl_up(v3, 3, 64);
            const float sum = v0 + (up[1] ? u1 : 0.f) + (up[2] ? u2 : 0.f) + (up[3] ? u3 : 0.f);
            if (!rowok) continue;
            if (t.x0 >= 0 && t.x0 < W && sum != 0.f) flow_sink<TILED>(win, gp, c, P, W, yy, t.x0, sum);
            if (!dn[1] && v1 != 0.f && t.x0 + 1 >= 0 && t.x0 + 1 < W) flow_sink<TILED>(win, gp, c, P, W, yy, t.x0 + 1, v1);
            if (!dn[2] && v2 != 0.f && t.x0 + 2 >= 0 && t.x0 + 2 < W) flow_sink<TILED>(win, gp, c, P, W, yy, t.x0 + 2, v2);
            if (!dn[3] && v3 != 0.f && t.x0 + 3 >= 0 && t.x0 + 3 < W) flow_sink<TILED>(win, gp, c, P, W, yy, t.x0 + 3, v3);
        }
    }
    return s;
}
template <bool TILED>
__global__ __launch_bounds__(256, 4) void k_flow_loss(const float* __restrict__ cat, const int* __restrict__ idx, const float* __restrict__ flows,
                                                   const float* __restrict__ masks, int b, int H, int W, float scale, float* __restrict__ gimg,
                                                   fxq_t* __restrict__ gpre, fx_t* __restrict__ acc, int tiles_x, const int* __restrict__ flow_shift) {
    __shared__ float red[16];
    __shared__ int wmin[2];
    extern __shared__ __attribute__((aligned(16))) int fwin[];
    const int j = blockIdx.y, f = idx[j], P = H * W;
    if (f == 0) return;  // valid = idx > 0
    const float* img = cat + (size_t)j * 3 * P; const float* pre = cat + (size_t)(b + j) * 3 * P;
    float* gi = gimg + (size_t)j * 3 * P; fxq_t* gp = gpre + (size_t)j * 3 * P;
    const float* fl = flows + (size_t)f * 2 * P; const float* mk = masks + (size_t)f * P;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    FlowWin win = {fwin, 0, 0, __builtin_ldexpf(1.f, flow_shift[f])};
    if (!TILED) {
        // every lane of a wave runs the same trip count (the shuffles need all 64 lanes); lanes past the image are `live == false`
        for (int p0 = blockIdx.x * blockDim.x; p0 < P; p0 += gridDim.x * blockDim.x) {
            const int p = p0 + threadIdx.x;
            const bool live = p < P;
            const int y = (live ? p : P - 1) / W, x = (live ? p : P - 1) - y * W;
            s += flow_pixel<false>(img, pre, fl, mk, gi, gp, win, x, y, live, lane, H, W, scale);
        }
    } else {
        const int X0 = ((int)blockIdx.x % tiles_x) * FT_W, Y0 = ((int)blockIdx.x / tiles_x) * FT_H, wv = threadIdx.x >> 6;
        for (int i = threadIdx.x; i < 3 * FT_WY * FT_WX; i += 256) fwin[i] = 0;
        if (threadIdx.x < 2) wmin[threadIdx.x] = 0x7fffffff;
        __syncthreads();
        int mx = 0x7fffffff, my = 0x7fffffff;                     // minimum tap origin of the tile's live pixels: the window's corner
#pragma unroll
        for (int r = 0; r < FT_H / 4; ++r) {
            const int x = X0 + lane, y = Y0 + r * 4 + wv;
            if (x < W && y < H) { Tap t = make_tap(fl[y * W + x], fl[P + y * W + x], x, y, W, H); mx = min(mx, t.x0); my = min(my, t.y0); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx = min(mx, __shfl_xor(mx, o, 64)); my = min(my, __shfl_xor(my, o, 64)); }
        if (lane == 0) { atomicMin(&wmin[0], mx); atomicMin(&wmin[1], my); }
        __syncthreads();
        win.x0 = wmin[0]; win.y0 = wmin[1];
#pragma unroll 1
        for (int r = 0; r < FT_H / 4; ++r) {
            const int x = X0 + lane, y = Y0 + r * 4 + wv;
            s += flow_pixel<true>(img, pre, fl, mk, gi, gp, win, x, y, x < W && y < H, lane, H, W, scale);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * FT_WY * FT_WX; i += 256) {
            const int v = fwin[i];
            if (v) {
                const int c = i / (FT_WY * FT_WX), rem = i - c * (FT_WY * FT_WX), ly = rem / FT_WX, lx = rem - ly * FT_WX;
                atomicAdd(gp + (size_t)c * P + (size_t)(win.y0 + ly) * W + win.x0 + lx, v);
            }
        }
    }
    float r = block_sum(s, red);
    if (threadIdx.x == 0) fx_add(acc + ((blockIdx.x + blockIdx.y) & (ACC_SLOTS - 1)) * 4 + 3, r, FX_ACC);
}
// ---- per-frame cell load of the flow scatter -> flow_shift[f] (FX_FLOW_SHIFT comment).  A pixel with tap origin (x0, y0) touches the cells
// [x0, x0 + 3] x [y0, y0 + 3], so the number of masked-in pixels reaching cell (cx, cy) is the 4x4 box sum of the ORIGIN histogram over
// [cx - 3, cx] x [cy - 3, cy]: one integer atomic per pixel, then one box sum per cell.  Origins left of / above the image are clamped to 0
// (their window then still covers every in-image cell they touch: the bound only grows).
__global__ void k_flow_origin_hist(const float* __restrict__ fl, const float* __restrict__ mk, int H, int W, int* __restrict__ hist) {
    const int P = H * W;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        if (!(mk[p] != 0.f)) continue;
        const int y = p / W, x = p - y * W;
        const Tap t = make_tap(fl[p], fl[P + p], x, y, W, H);
        if (t.x0 >= W || t.y0 >= H || t.x0 + 3 < 0 || t.y0 + 3 < 0) continue;      // no tap inside the image
        atomicAdd(hist + max(t.y0, 0) * W + max(t.x0, 0), 1);
    }
}
__global__ void k_flow_cell_load(int* __restrict__ hist_max, const int* __restrict__ hist, int H, int W) {
    __shared__ int red[4];
    int mx = 0;
    const int P = H * W;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int cy = p / W, cx = p - cy * W;
        int s = 0;
        for (int j = 0; j < 4; ++j) { const int yy = cy - j; if (yy < 0) break;
            for (int i = 0; i < 4; ++i) { const int xx = cx - i; if (xx < 0) break; s += hist[yy * W + xx]; } }
        mx = max(mx, s);
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(hist_max, max(max(red[0], red[1]), max(red[2], red[3])));
}
__global__ void k_flow_shift_set(const int* __restrict__ load, int* __restrict__ shift) {
    const int L = max(*load, 1);
    int lg = 0; while ((1 << lg) < L) ++lg;                 // ceil(log2 L)
    *shift = min(FX_FLOW_SHIFT, 30 - lg);
}
static int g_flow_tiled = -1;
static void launch_flow_loss(const float* cat, const int* cidx, const float* flows, const float* masks, int b, int H, int W, float fscale, float* gimg,
                             fxq_t* gpre, fx_t* acc, dim3 untiled_grid, const int* flow_shift, hipStream_t st) {
    if (g_flow_tiled < 0) g_flow_tiled = getenv("TCL_FLOW_TILED") ? atoi(getenv("TCL_FLOW_TILED")) : 1;      // A/B hook: 0 = global atomics only
    if (g_flow_tiled) {
        const int tx = cdiv(W, FT_W), ty = cdiv(H, FT_H);
        hipLaunchKernelGGL(k_flow_loss<true>, dim3(tx * ty, b), dim3(256), (size_t)3 * FT_WY * FT_WX * 4, st, cat, cidx, flows, masks, b, H, W, fscale, gimg, gpre, acc, tx, flow_shift);
    } else hipLaunchKernelGGL(k_flow_loss<false>, untiled_grid, dim3(256), 0, st, cat, cidx, flows, masks, b, H, W, fscale, gimg, gpre, acc, 0, flow_shift);
}
// loss = w_photo*(c_l1*acc0 + msssim_term) + w_flow*acc3/cnt_flow + tv ; acc reset for the next iteration
__global__ void k_loss_finalize(fx_t* acc, const float* ms_term, float w_photo, float c_l1, float w_flow, float inv_cnt_flow,
                                float c_tvh, float c_tvw, float* loss_out) {
    fx_t t[4] = {0, 0, 0, 0};
    for (int sidx = 0; sidx < ACC_SLOTS; ++sidx)
        for (int k = 0; k < 4; ++k) { t[k] += acc[sidx * 4 + k]; acc[sidx * 4 + k] = 0; }
    float a[4];
    for (int k = 0; k < 4; ++k) a[k] = fx_get(t[k], 1.f / FX_ACC);
    *loss_out = w_photo * (c_l1 * a[0] + *ms_term) + w_flow * a[3] * inv_cnt_flow + c_tvh * a[1] + c_tvw * a[2];
}

// ---------------------------------------------------------------- optimiser / init
// torch.optim.Adam, one element.  ONE function for the dense step, the lazy step and the replay of skipped steps (g = 0): the same
// instruction sequence, so the lazy schedule below reproduces the dense one bit for bit.
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    // no fma contraction in here: the compiler is otherwise free to fuse b1*m + (1-b1)*g one way in one kernel and the other way in another (or
    // to fold the g = 0 replay differently) -- a 1-ulp difference that Adam(eps 1e-15) turns into a +-lr step wherever momentum and gradient
    // nearly cancel (measured: dense vs lazy 1 ulp apart after 2 iterations, 5e-3 after 25).  (HIP's __fmul_rn & co are plain operators.)
#pragma clang fp contract(off)
    const float mi = b1 * m + (1.f - b1) * g, vi = b2 * v + (1.f - b2) * g * g;
    m = mi; v = vi;
    p -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}
// dense step; g is zeroed for the next iteration (saves a memset pass).
__global__ void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                       float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, mi, vi, g[i], lr, b1, b2, eps, bc1, bc2_sqrt);
        p[i] = pi; m[i] = mi; v[i] = vi; g[i] = 0.f;
    }
}
// ---- LAZY dense Adam (stage 2, round 3).  The reference's Adam is dense: every codebook row moves every iteration through its momentum,
// 84 B per row and iteration (p, g, m, v read; p, m, v written) -- at K ~ N H W rows (short tracks) that stream is 9/10 of the stage's
// traffic, although only the rows of the mini-batch's 2 b frames receive a gradient.  A row's update in an iteration WITHOUT gradient is a
// pure function of its own (p, m, v) and the step number, so it can be applied later: t_last[row] = the last step applied; before a row is
// gathered (it is in the mini-batch) the steps t_last+1 .. it are replayed with g = 0 by the SAME instruction sequence the dense kernel runs
// (adam_elem), then the step with its gradient follows, and one dense catch-up pass ends the stage.  Bit-identical to the dense schedule
// (tests/test_gpu_path2.py::test_stage2_lazy_adam_equals_dense); traffic per iteration ~ the mini-batch's rows instead of all K.
// Rows are visited frame by frame (one launch per cat row, like k_codebook_bwd<false>): ids are distinct inside a frame, and a row shared by
// two frames of the batch is handled by whichever launch comes first (the t_last test).
// t_last[row] = the last step applied to the row, plus (round 5, second pass) bit 30 = the row is SHARED: more than one pixel of the clip maps to it, so two
// frames of a mini-batch may reach it in one launch and it must be claimed with an atomic.  A row that a single pixel holds (a track of length one: 98 % of the
// rows with the bench clip's ids) has nobody to race with: its counter is read and written with plain accesses.  Every pixel's atomic was what bound the two
// visiting kernels -- ~17 G device-scope atomics per second on scattered addresses -- more than their 108 bytes per row.
// The flags are built once per stage without atomics: every pixel writes its own number into a scratch word of its row (plane 0 of the still-zero gradient),
// then every pixel looks whether its number survived; a pixel that finds another one marks the row (idempotent plain store).
#define TL_SHARED 0x40000000
#define TL_STEP(x) ((x) & 0x3fffffff)
__global__ void k_tl_mark(const int* __restrict__ inv, size_t total, int* __restrict__ scratch) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) scratch[inv[i]] = (int)i + 1;
}
__global__ void k_tl_flag(const int* __restrict__ inv, size_t total, const int* __restrict__ scratch, int* __restrict__ t_last) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int id = inv[i];
        if (scratch[id] != (int)i + 1) t_last[id] = TL_SHARED;
    }
}
__device__ __forceinline__ void adam_replay(float (&p)[3], float (&m)[3], float (&v)[3], int from, int to, float lr, float b1, float b2, float eps,
                                            const float* __restrict__ bc1, const float* __restrict__ bc2) {
    if (m[0] == 0.f && m[1] == 0.f && m[2] == 0.f && v[0] == 0.f && v[1] == 0.f && v[2] == 0.f) return;      // never touched: every skipped step is a no-op
    for (int s = from; s <= to; ++s) {
        const float c1 = bc1[s], c2 = bc2[s];
#pragma unroll
        for (int c = 0; c < 3; ++c) adam_elem(p[c], m[c], v[c], 0.f, lr, b1, b2, eps, c1, c2);
    }
}
// rows of the cat rows blockIdx.y (all frames of a mini-batch in ONE launch): bring them to step `upto` (no gradient).  A track that several
// frames of the batch share is claimed by exactly one thread (atomicMax on its step counter returns the old value to the first comer only);
// whoever wins computes the same values, nobody else touches the row in this launch, and its readers are later kernels: same bits as the
// frame-by-frame launches this replaces (32 launches of ~34 us per iteration at batch 16).
__global__ void k_adam_catchup_frame(const int* __restrict__ inv, const int* __restrict__ fidx, int P, size_t K, int* __restrict__ t_last,
                                     float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int upto, float lr, float b1, float b2,
                                     float eps, const float* __restrict__ bc1, const float* __restrict__ bc2) {
    // a frame can sit in the cat list twice (as a current frame and as the previous frame of another slot): only its FIRST occurrence walks its pixels --
    // with the unshared rows no longer claimed by an atomic, two blocks on the same pixel would both step its row
    for (int q = 0; q < (int)blockIdx.y; ++q) if (fidx[q] == fidx[blockIdx.y]) return;
    const int* iv = inv + (size_t)fidx[blockIdx.y] * P;
    for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < P; px += gridDim.x * blockDim.x) {
        const size_t id = (size_t)iv[px];
        const int raw = t_last[id];
        if (TL_STEP(raw) >= upto) continue;
        int tl = TL_STEP(raw);
        if (raw & TL_SHARED) {                      // a track that other pixels of the clip share: claim it (one winner per launch)
            tl = TL_STEP(atomicMax(t_last + id, TL_SHARED | upto));
            if (tl >= upto) continue;
        } else {                                    // a row only this pixel holds: nobody to race with, no atomic (98 % of the rows in the bench's regime)
            t_last[id] = upto;
            if (tl == 0) continue;                  // ... and never stepped: m = v = 0, every skipped step is a no-op -- no row traffic at all (round 6)
        }
        float pp[3], mm[3], vv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pp[c] = p[c * K + id]; mm[c] = m[c * K + id]; vv[c] = v[c * K + id]; }
        adam_replay(pp, mm, vv, tl + 1, upto, lr, b1, b2, eps, bc1, bc2);
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c * K + id] = pp[c]; m[c * K + id] = mm[c]; v[c * K + id] = vv[c]; }
    }
}
// rows of the cat rows blockIdx.y that stand at step - 1: apply `step` with their (complete) gradient, clear it.  One thread per row wins the
// compare-and-swap of the step counter.
// Round 6 -- RUN-AHEAD for unshared rows.  The mini-batch schedule of the whole stage is known on the host, so for every cat row of an iteration the NEXT
// iteration that holds the same frame is known (`ahead.v[j]`: the step count the row must stand at when it is next gathered; the stage's last step if it
// never returns).  A row that a single pixel holds is only ever visited through that frame: after its gradient step it is replayed (g = 0, the dense
// kernel's instruction sequence, adam_elem) straight to that step while its p, m, v sit in registers, and t_last says so.  The catch-up launch of its next
// visit then finds it up to date: 9 loads + 9 stores per row and visit less (2.6 GB of the 15 GB an iteration moved), the replay arithmetic is done once
// as before, now under this kernel's own memory traffic instead of in a launch of its own.  Shared rows (TL_SHARED: several pixels, possibly of frames
// with different schedules) keep the visit-time catch-up.  Same bits as the dense schedule (test_stage2_lazy_adam_equals_dense).
struct AheadArg { int v[128]; };
__global__ void k_adam_touched_frame(const int* __restrict__ inv, const int* __restrict__ fidx, int P, size_t K, int* __restrict__ t_last,
                                     float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int step, float lr,
                                     float b1, float b2, float eps, float bc1, float bc2_sqrt, const float* __restrict__ bc1_tab,
                                     const float* __restrict__ bc2_tab, AheadArg ahead) {
    const int upto_next = ahead.v[blockIdx.y];      // >= step; == step: no run-ahead (TCL_ADAM_RUNAHEAD=0)
    // a frame can sit in the cat list twice (as a current frame and as the previous frame of another slot): only its FIRST occurrence walks its pixels --
    // with the unshared rows no longer claimed by an atomic, two blocks on the same pixel would both step its row
    for (int q = 0; q < (int)blockIdx.y; ++q) if (fidx[q] == fidx[blockIdx.y]) return;
    const int* iv = inv + (size_t)fidx[blockIdx.y] * P;
    for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < P; px += gridDim.x * blockDim.x) {
        const size_t id = (size_t)iv[px];
        const int raw = t_last[id];
        if (TL_STEP(raw) != step - 1) continue;
        const bool shared = raw & TL_SHARED;
        if (shared) {
            if (atomicCAS(t_last + id, TL_SHARED | (step - 1), TL_SHARED | step) != (TL_SHARED | (step - 1))) continue;      // stepped by another frame of this mini-batch
        } else t_last[id] = upto_next;              // unshared row: no claim needed; it leaves this kernel at the step of its next visit
        float pp[3], mm[3], vv[3], gg[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pp[c] = p[c * K + id]; mm[c] = m[c * K + id]; vv[c] = v[c * K + id]; gg[c] = g[c * K + id]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) adam_elem(pp[c], mm[c], vv[c], gg[c], lr, b1, b2, eps, bc1, bc2_sqrt);
        if (!shared && upto_next > step) adam_replay(pp, mm, vv, step + 1, upto_next, lr, b1, b2, eps, bc1_tab, bc2_tab);
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c * K + id] = pp[c]; m[c * K + id] = mm[c]; v[c * K + id] = vv[c]; g[c * K + id] = 0.f; }
    }
}
// ---- round 5: ONE visit per row and iteration with a write.  The catch-up launch is gone: the gather replays a row's skipped steps in REGISTERS
// (read-only: p, m, v, t_last) and hands the clamp mask on; the step kernel replays them again (the same instruction sequence: same bits) and
// applies the gradient.  Per row 156 B instead of 208 B of traffic, one launch less per iteration.
__global__ void k_gather_codebook_lazy(const float* __restrict__ feat, const float* __restrict__ m, const float* __restrict__ v,
                                       const int* __restrict__ t_last, const int* __restrict__ inv, const int* __restrict__ fidx,
                                       float* __restrict__ out, unsigned char* __restrict__ cmask, int P, size_t K, int upto, float lr, float b1,
                                       float b2, float eps, const float* __restrict__ bc1, const float* __restrict__ bc2) {
    const int j = blockIdx.y, f = fidx[j];
    const int* iv = inv + (size_t)f * P; float* o = out + (size_t)j * 3 * P; unsigned char* cm = cmask + (size_t)j * P;
    for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < P; px += gridDim.x * blockDim.x) {
        const size_t id = (size_t)iv[px];
        const int tl = TL_STEP(t_last[id]);
        float pp[3], mm[3], vv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pp[c] = feat[c * K + id];
        if (tl < upto) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { mm[c] = m[c * K + id]; vv[c] = v[c * K + id]; }
            adam_replay(pp, mm, vv, tl + 1, upto, lr, b1, b2, eps, bc1, bc2);
        }
        int mk = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float val = pp[c] * SH_C0 + 0.5f;
            mk |= (val >= 0.f && val <= 1.f) ? (1 << c) : 0;
            o[c * P + px] = fminf(fmaxf(val, 0.f), 1.f);
        }
        cm[px] = (unsigned char)mk;
    }
}
// rows of the cat rows blockIdx.y: whatever step they stand at (< step), replay up to step - 1 without gradient, then apply `step` with the
// (complete) gradient and clear it.  One thread per row wins the atomicMax of the step counter and learns where the row stood.
__global__ void k_adam_step_rows_lazy(const int* __restrict__ inv, const int* __restrict__ fidx, int P, size_t K, int* __restrict__ t_last,
                                      float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int step, float lr,
                                      float b1, float b2, float eps, const float* __restrict__ bc1, const float* __restrict__ bc2) {
    // a frame can sit in the cat list twice (as a current frame and as the previous frame of another slot): only its FIRST occurrence walks its pixels --
    // with the unshared rows no longer claimed by an atomic, two blocks on the same pixel would both step its row
    for (int q = 0; q < (int)blockIdx.y; ++q) if (fidx[q] == fidx[blockIdx.y]) return;
    const int* iv = inv + (size_t)fidx[blockIdx.y] * P;
    const float c1 = bc1[step], c2 = bc2[step];
    for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < P; px += gridDim.x * blockDim.x) {
        const size_t id = (size_t)iv[px];
        const int raw = t_last[id];
        if (TL_STEP(raw) >= step) continue;
        int tl = TL_STEP(raw);
        if (raw & TL_SHARED) {
            tl = TL_STEP(atomicMax(t_last + id, TL_SHARED | step));
            if (tl >= step) continue;                              // stepped by another frame of this mini-batch
        } else t_last[id] = step;
        float pp[3], mm[3], vv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pp[c] = p[c * K + id]; mm[c] = m[c * K + id]; vv[c] = v[c * K + id]; }
        if (tl < step - 1) adam_replay(pp, mm, vv, tl + 1, step - 1, lr, b1, b2, eps, bc1, bc2);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            adam_elem(pp[c], mm[c], vv[c], g[c * K + id], lr, b1, b2, eps, c1, c2);
            p[c * K + id] = pp[c]; m[c * K + id] = mm[c]; v[c * K + id] = vv[c]; g[c * K + id] = 0.f;
        }
    }
}
// end of the stage: every row to the last step
__global__ void k_adam_catchup_all(size_t K, int* __restrict__ t_last, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int upto,
                                   float lr, float b1, float b2, float eps, const float* __restrict__ bc1, const float* __restrict__ bc2) {
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < K; id += (size_t)gridDim.x * blockDim.x) {
        const int tl = TL_STEP(t_last[id]);
        if (tl >= upto) continue;
        float pp[3], mm[3], vv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pp[c] = p[c * K + id]; mm[c] = m[c * K + id]; vv[c] = v[c * K + id]; }
        adam_replay(pp, mm, vv, tl + 1, upto, lr, b1, b2, eps, bc1, bc2);
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c * K + id] = pp[c]; m[c * K + id] = mm[c]; v[c * K + id] = vv[c]; }
        t_last[id] = upto;
    }
}
template <bool ATOMIC>
__global__ void k_scatter_accum(const float* __restrict__ img, const int* __restrict__ inv, float* __restrict__ sum,
                                float* __restrict__ cnt, int P, size_t K, int f0) {
    const int f = f0 + blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        size_t id = inv[(size_t)f * P + p];
        const float* s = img + (size_t)f * 3 * P;
        if (ATOMIC) { atomicAdd(sum + id, s[p]); atomicAdd(sum + K + id, s[P + p]); atomicAdd(sum + 2 * K + id, s[2 * P + p]); atomicAdd(cnt + id, 1.f); }
        else { sum[id] += s[p]; sum[K + id] += s[P + p]; sum[2 * K + id] += s[2 * P + p]; cnt[id] += 1.f; }     // ids distinct within frame f
    }
}
__global__ void k_scatter_final(float* __restrict__ feat, const float* __restrict__ cnt, size_t K) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < K * 3; i += (size_t)gridDim.x * blockDim.x)
        feat[i] = (feat[i] / fmaxf(cnt[i % K], 1.f) - 0.5f) / SH_C0;  // mean then RGB2SH (generate.py:478-479); planar [3,K]
}

// ================================================================= host side (C ABI)
static inline dim3 pgrid(int P, int ny) { return dim3(stream_grid(P, 256, 4) > 1024 ? 1024 : stream_grid(P, 256, 4), ny); }
static inline int pooled(int s) { return (s + 2 * (s & 1) - 2) / 2 + 1; }

extern "C" {

int tcl_flow_scatter_mode(int tiled) { g_flow_tiled = tiled; return TCL_OK; }

int tcl_flow_cell_shift(const float* flows, const float* masks, int N, int H, int W, int* scratch, int* flow_shift, hipStream_t st) {
    TCL_CHECK_ARG(flows && masks && scratch && flow_shift && N > 0 && H > 3 && W > 3);
    const size_t P = (size_t)H * W;
    for (int f = 0; f < N; ++f) {                             // (frame 0 is never a flow target: its entry is computed like any other)
        if (hipMemsetAsync(scratch, 0, (P + 1) * 4, st) != hipSuccess) return TCL_ELAUNCH;
        hipLaunchKernelGGL(k_flow_origin_hist, pgrid(P, 1), dim3(256), 0, st, flows + (size_t)f * 2 * P, masks + (size_t)f * P, H, W, scratch);
        hipLaunchKernelGGL(k_flow_cell_load, pgrid(P, 1), dim3(256), 0, st, scratch + P, scratch, H, W);
        hipLaunchKernelGGL(k_flow_shift_set, dim3(1), dim3(1), 0, st, scratch + P, flow_shift + f);
    }
    TCL_LAUNCH_RET();
}


int tcl_warp_flow_fwd(const float* img, const float* flow, float* out, int n, int c, int h, int w, int flow_c, hipStream_t st) {
    TCL_CHECK_ARG(img && flow && out && n > 0 && c > 0 && h > 1 && w > 1 && flow_c >= 2);
    hipLaunchKernelGGL(k_warp_fwd, pgrid(h * w, n), dim3(256), 0, st, img, flow, out, c, h, w, flow_c);
    TCL_LAUNCH_RET();
}
int tcl_warp_flow_bwd(const float* gout, const float* flow, float* gimg, int n, int c, int h, int w, int flow_c, hipStream_t st) {
    TCL_CHECK_ARG(gout && flow && gimg && n > 0 && c > 0 && h > 1 && w > 1 && flow_c >= 2);
    if (hipMemsetAsync(gimg, 0, (size_t)n * c * h * w * 4, st) != hipSuccess) return TCL_ELAUNCH;
    hipLaunchKernelGGL(k_warp_bwd, pgrid(h * w, n), dim3(256), 0, st, gout, flow, gimg, c, h, w, flow_c);
    TCL_LAUNCH_RET();
}
int tcl_apply_exposure(const float* src, const int* idx, const float* expo, float* out, int nb, int h, int w, hipStream_t st) {
    TCL_CHECK_ARG(src && expo && out && nb > 0);
    hipLaunchKernelGGL(k_apply_exposure, pgrid(h * w, nb), dim3(256), 0, st, src, idx, expo, out, h * w);
    TCL_LAUNCH_RET();
}
int tcl_gather_codebook(const float* feat, const int* inv, const int* fidx, float* out, int nb, int h, int w, size_t K, hipStream_t st) {
    TCL_CHECK_ARG(feat && inv && out && nb > 0);
    hipLaunchKernelGGL(k_gather_codebook, pgrid(h * w, nb), dim3(256), 0, st, feat, inv, fidx, out, h * w, K, (unsigned char*)nullptr);
    TCL_LAUNCH_RET();
}
int tcl_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step, hipStream_t st) {
    TCL_CHECK_ARG(p && g && m && v && step >= 1);
    float bc1 = (float)(1.0 - pow((double)b1, step)), bc2 = (float)sqrt(1.0 - pow((double)b2, step));
    hipLaunchKernelGGL(k_adam, dim3(stream_grid((long)n, 256, 4)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2);
    TCL_LAUNCH_RET();
}
int tcl_scatter_mean_rgb2sh(const float* img, const int* inv, float* feat, float* cnt, int n, int h, int w, size_t K, int ids_unique, hipStream_t st) {
    TCL_CHECK_ARG(img && inv && feat && cnt && K > 0);
    if (hipMemsetAsync(feat, 0, K * 12, st) != hipSuccess || hipMemsetAsync(cnt, 0, K * 4, st) != hipSuccess) return TCL_ELAUNCH;
    if (ids_unique)          // frame after frame: conflict-free inside a frame, fixed order across frames (bit-reproducible sums)
        for (int f = 0; f < n; ++f) hipLaunchKernelGGL(k_scatter_accum<false>, pgrid(h * w, 1), dim3(256), 0, st, img, inv, feat, cnt, h * w, K, f);
    else hipLaunchKernelGGL(k_scatter_accum<true>, pgrid(h * w, n), dim3(256), 0, st, img, inv, feat, cnt, h * w, K, 0);
    hipLaunchKernelGGL(k_scatter_final, dim3(stream_grid((long)K * 3, 256, 4)), dim3(256), 0, st, feat, cnt, K);
    TCL_LAUNCH_RET();
}
// Small host tables reach the device as KERNEL ARGUMENTS (copied at launch time): no pageable-memory hipMemcpyAsync whose source must outlive
// the call, no host synchronisation inside the library.
struct TabChunk { float v[512]; };
__global__ void k_fill_table(float* dst, int n, TabChunk c) { const int i = threadIdx.x; if (i < n) dst[i] = c.v[i]; }
__global__ void k_set_i32(int* dst, int v) { *dst = v; }
static int upload_table(float* dst, const float* src, size_t n, hipStream_t st) {
    for (size_t o = 0; o < n; o += 512) {
        TabChunk c;
        const int m = (int)(n - o < 512 ? n - o : 512);
        for (int i = 0; i < m; ++i) c.v[i] = src[o + i];
        hipLaunchKernelGGL(k_fill_table, dim3(1), dim3(512), 0, st, dst + o, m, c);
    }
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}
int tcl_track_ids_unique(const int* unq_inv, int N, int H, int W, size_t K, int* scratch, int* result, hipStream_t st) {
    TCL_CHECK_ARG(unq_inv && scratch && result && N > 0 && H > 0 && W > 0 && K > 0);
    const int P = H * W;
    hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, st, result, 1);
    for (int f = 0; f < N; ++f) {
        hipLaunchKernelGGL(k_ids_mark, pgrid(P, 1), dim3(256), 0, st, unq_inv + (size_t)f * P, P, scratch);
        hipLaunchKernelGGL(k_ids_check, pgrid(P, 1), dim3(256), 0, st, unq_inv + (size_t)f * P, P, scratch, result);
    }
    TCL_LAUNCH_RET();
}

// workspace carve for the MS-SSIM chain over `planes` planes of h x w
struct MsWs { float *X[5], *Y[5], *mA[5], *mB[5], *mC[5], *gX[5], *scal, *cnt, *term; fx_t* sums; int h[5], w[5]; size_t bytes; };
static MsWs carve_ms(char* base, int planes, int h, int w) {
    MsWs W; memset(&W, 0, sizeof(W));
    size_t off = 0;
    auto take = [&](size_t nfloat) { float* p = (float*)(base ? base + off : nullptr); off += (nfloat * 4 + 255) & ~(size_t)255; return p; };
    W.h[0] = h; W.w[0] = w;
    for (int l = 1; l < 5; ++l) { W.h[l] = pooled(W.h[l - 1]); W.w[l] = pooled(W.w[l - 1]); }
    for (int l = 1; l < 5; ++l) {
        size_t n = (size_t)planes * W.h[l] * W.w[l], no = (size_t)planes * (W.h[l] - HALO) * (W.w[l] - HALO);
        W.X[l] = take(n); W.Y[l] = take(n); W.gX[l] = take(n); W.mA[l] = take(no); W.mB[l] = take(no); W.mC[l] = take(no);
    }
    W.sums = (fx_t*)take((size_t)8 * planes); W.scal = take((size_t)4 * planes); W.cnt = take(4); W.term = take(1);
    W.bytes = off;
    return W;
}
size_t tcl_msssim_workspace_bytes(int planes, int h, int w) { return carve_ms(nullptr, planes, h, w).bytes + 256; }

// forward (+ optional backward to level-1 grad) of lambda*(1 - relaxed_ms_ssim(X, Y, start_level=1)).
// X: [planes] contiguous planes; Y planes addressed through yidx (frame ids, 3 planes per frame) or contiguous.
static int msssim_chain(const float* X, const float* Y, const int* yidx, int planes, int h, int w, float lambda, MsWs& W,
                        bool backward, hipStream_t st, int planes_norm = 0) {
    if (planes_norm <= 0) planes_norm = planes;
    static const Gauss11 G = make_gauss();
    for (int l = 1; l < 5; ++l) if (W.h[l] < 11 || W.w[l] < 11) return TCL_EINVAL;
    Cnt4 cn;
    for (int l = 1; l < 5; ++l) cn.c[l - 1] = (float)(W.h[l] - HALO) * (float)(W.w[l] - HALO);
    if (hipMemsetAsync(W.sums, 0, (size_t)4 * planes * sizeof(fx_t), st) != hipSuccess) return TCL_ELAUNCH;
    for (int l = 1; l < 5; ++l) {
        dim3 g(cdiv((long)W.h[l] * W.w[l], 256) > 512 ? 512 : cdiv((long)W.h[l] * W.w[l], 256), planes);
        hipLaunchKernelGGL(k_pool2, g, dim3(256), 0, st, l == 1 ? X : W.X[l - 1], (const int*)nullptr, W.X[l], W.h[l - 1], W.w[l - 1], W.h[l], W.w[l]);
        hipLaunchKernelGGL(k_pool2, g, dim3(256), 0, st, l == 1 ? Y : W.Y[l - 1], l == 1 ? yidx : (const int*)nullptr, W.Y[l], W.h[l - 1], W.w[l - 1], W.h[l], W.w[l]);
        dim3 gs(cdiv(W.w[l] - HALO, TW), cdiv(W.h[l] - HALO, TH), planes);
        hipLaunchKernelGGL(k_ssim_fwd, gs, dim3(256), 0, st, W.X[l], W.Y[l], W.h[l], W.w[l], l == 4 ? 1 : 0, 0.0001f, 0.0009f, G,
                           W.mA[l], W.mB[l], W.mC[l], W.sums + (size_t)(l - 1) * planes);
    }
    hipLaunchKernelGGL(k_msssim_finalize, dim3(1), dim3(256), 0, st, W.sums, planes, planes_norm, cn, lambda, W.scal, W.term);
    if (backward)
        for (int l = 4; l >= 1; --l) {
            dim3 gs(cdiv(W.w[l], TW), cdiv(W.h[l], TH), planes);
            hipLaunchKernelGGL(k_ssim_bwd, gs, dim3(256), 0, st, W.X[l], W.Y[l], W.h[l], W.w[l], W.mA[l], W.mB[l], W.mC[l],
                               W.scal + (size_t)(l - 1) * planes, G, l < 4 ? W.gX[l + 1] : (const float*)nullptr,
                               l < 4 ? W.h[l + 1] : 0, l < 4 ? W.w[l + 1] : 0, W.gX[l]);
        }
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

// value = 1 - relaxed_ms_ssim(X, Y, data_range=1, start_level=1); gradX = d(value)/dX (may be null)
int tcl_ms_ssim_loss(const float* X, const float* Y, int planes, int h, int w, float* value, float* gradX, void* ws, hipStream_t st) {
    TCL_CHECK_ARG(X && Y && value && ws && planes > 0);
    MsWs W = carve_ms((char*)ws, planes, h, w);
    int rc = msssim_chain(X, Y, nullptr, planes, h, w, 1.f, W, gradX != nullptr, st);
    if (rc) return rc;
    if (gradX) {
        hipLaunchKernelGGL(k_pixel_losses, pgrid(h * w, planes), dim3(256), 0, st, X, (const float*)nullptr, (const int*)nullptr, W.gX[1],
                           W.h[1], W.w[1], h, w, 0.f, 0.f, 0.f, gradX, (fx_t*)nullptr);
    }
    if (hipMemcpyAsync(value, W.term, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return TCL_ELAUNCH;
    TCL_LAUNCH_RET();
}

int tcl_tv_loss(const float* x, int b, int c, int h, int w, float weight, float* value, float* grad, void* ws16, hipStream_t st) {
    TCL_CHECK_ARG(x && value && grad && ws16 && b > 0);
    fx_t* acc = (fx_t*)ws16;                                                  // [ACC_SLOTS][4] fixed-point cells, then one zero float
    float* zero = (float*)(acc + ACC_SLOTS * 4);
    if (hipMemsetAsync(acc, 0, ACC_SLOTS * 4 * sizeof(fx_t) + 4, st) != hipSuccess) return TCL_ELAUNCH;
    float ch = weight * 2.f / ((float)c * (h - 1) * w) / b, cw = weight * 2.f / ((float)c * h * (w - 1)) / b;
    hipLaunchKernelGGL(k_pixel_losses, pgrid(h * w, b * c), dim3(256), 0, st, x, (const float*)nullptr, (const int*)nullptr,
                       (const float*)nullptr, 0, 0, h, w, 0.f, ch, cw, grad, acc);
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1), 0, st, acc, zero, 0.f, 0.f, 0.f, 0.f, ch, cw, value);
    TCL_LAUNCH_RET();
}

// ---- whole-stage drivers -------------------------------------------------------------------
struct StageWs { float *cat, *gimg; fxq_t* gpre; fx_t *acc, *efx; int* cidx; unsigned char* cmask; MsWs ms; size_t bytes; };
static StageWs carve_stage(char* base, int b, int h, int w) {
    StageWs S; size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    size_t P = (size_t)h * w;
    S.cat = (float*)take(2 * b * 3 * P * 4); S.gimg = (float*)take(b * 3 * P * 4); S.gpre = (fxq_t*)take(b * 3 * P * sizeof(fxq_t));
    S.acc = (fx_t*)take(ACC_SLOTS * 4 * sizeof(fx_t)); S.efx = (fx_t*)take((size_t)2 * b * 12 * sizeof(fx_t));
    S.cidx = (int*)take(2 * b * 4);
    S.cmask = (unsigned char*)take(2 * b * P);          // lazy stage 2: clamp mask of the gathered pixels (k_gather_codebook_lazy)
    size_t msb = carve_ms(nullptr, b * 3, h, w).bytes;
    char* mp = take(msb);
    S.ms = carve_ms(mp, b * 3, h, w);
    S.bytes = off;
    return S;
}
size_t tcl_stage_workspace_bytes(int batch, int h, int w) { return carve_stage(nullptr, batch, h, w).bytes + 256; }
size_t tcl_stage2_lazy_workspace_bytes(size_t K, int iters) { return ((K * 4 + 255) & ~(size_t)255) + (size_t)2 * (iters + 1) * 4 + 256; }

static double expon_lr(int step, double lr_init, double lr_final, int max_steps) {  // general_utils.py:31-64, delay off
    double t = (double)step / max_steps; t = t < 0 ? 0 : (t > 1 ? 1 : t);
    return exp(log(lr_init) * (1 - t) + log(lr_final) * t);
}

// ---- one mini-batch, gradient only.  The whole-stage drivers below and the multi-GPU host loop (tc_light_amd/post_opt.py: the batch's
// slots are dealt to the ranks, the gradients meet in a collective before the Adam step) share these.
// d_cidx (device) int32 [2*b_loc]: [cur(b_loc) | max(cur-1, 0)(b_loc)] -- THIS caller's slots of the mini-batch.  b_glob / nvalid_glob:
// slots and slots with idx > 0 of the WHOLE mini-batch: every mean of the loss (generate.py:413-427, :507-520) runs over the global batch,
// so partial losses / gradients of the ranks simply add up.  g is accumulated into (+=); *loss_part receives this caller's share of the loss.
int tcl_exposure_grad(const float* edited, const float* flows, const float* masks, const int* flow_shift, int N, int H, int W, const int* d_cidx, int b_loc,
                      int b_glob, int nvalid_glob, float lambda_dssim, float lambda_flow, const float* exposure, float* g,
                      float* loss_part, void* ws, hipStream_t st) {
    TCL_CHECK_ARG(edited && flows && masks && flow_shift && d_cidx && exposure && g && loss_part && ws);
    TCL_CHECK_ARG(N > 0 && b_loc > 0 && b_glob >= b_loc && nvalid_glob >= 0 && H > 160 && W > 160);
    StageWs S = carve_stage((char*)ws, b_loc, H, W);
    const size_t P = (size_t)H * W;
    const int b = b_loc;
    S.cidx = const_cast<int*>(d_cidx);
    if (hipMemsetAsync(S.acc, 0, ACC_SLOTS * 4 * sizeof(fx_t), st) != hipSuccess) return TCL_ELAUNCH;
    hipLaunchKernelGGL(k_apply_exposure, pgrid(P, 2 * b), dim3(256), 0, st, edited, S.cidx, exposure, S.cat, (int)P);
    const float wp = 1.f - lambda_flow, c_l1 = wp * (1.f - lambda_dssim) / ((float)b_glob * 3 * P);  // (1-lf) folded in
    int rc = msssim_chain(S.cat, edited, S.cidx, b * 3, H, W, wp * lambda_dssim, S.ms, true, st, b_glob * 3);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pixel_losses, pgrid(P, b * 3), dim3(256), 0, st, S.cat, edited, S.cidx, S.ms.gX[1], S.ms.h[1], S.ms.w[1], H, W,
                       c_l1, 0.f, 0.f, S.gimg, S.acc);
    if (hipMemsetAsync(S.gpre, 0, (size_t)b * 3 * P * sizeof(fxq_t), st) != hipSuccess) return TCL_ELAUNCH;
    if (hipMemsetAsync(S.efx, 0, (size_t)2 * b * 12 * sizeof(fx_t), st) != hipSuccess) return TCL_ELAUNCH;
    float inv_cnt = nvalid_glob ? 1.f / ((float)nvalid_glob * 3 * P) : 0.f;
    const float fscale = lambda_flow * inv_cnt;
    launch_flow_loss(S.cat, S.cidx, flows, masks, b, H, W, fscale, S.gimg, S.gpre, S.acc, pgrid(P, b), flow_shift, st);
    // 64 blocks per row (each block ends in 12 block-wide sums + 12 atomics: with the P / 1024-pixel blocks of the other kernels this was the
    // slowest kernel of stage 1), partial sums in fixed point, rows added to the gradient in order
    hipLaunchKernelGGL(k_exposure_bwd, dim3(64, 2 * b), dim3(256), 0, st, edited, S.cidx, exposure, S.gimg, S.gpre, fscale, flow_shift, b, S.efx, (int)P);
    hipLaunchKernelGGL(k_expo_fin, dim3(1), dim3(64), 0, st, S.efx, S.cidx, 2 * b, g);
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1), 0, st, S.acc, S.ms.term, 1.f, c_l1, lambda_flow, inv_cnt, 0.f, 0.f, loss_part);
    TCL_LAUNCH_RET();
}

struct LazyGather { const float *m, *v, *bc1, *bc2; const int* t_last; int upto; float lr; };      // the lazy schedule's gather (rows may be behind)
static int unique_tensor_grad_impl(const float* target, const float* flows, const float* masks, const int* flow_shift, const int* unq_inv, int N, int H, int W, size_t K,
                                   int ids_unique, const int* d_cidx, int b_loc, int b_glob, int nvalid_glob, float lambda_dssim, float lambda_flow,
                                   float lambda_tv, const float* feat, float* g, float* loss_part, void* ws, const LazyGather* lz, hipStream_t st);
int tcl_unique_tensor_grad(const float* target, const float* flows, const float* masks, const int* flow_shift, const int* unq_inv, int N, int H, int W, size_t K,
                           int ids_unique, const int* d_cidx, int b_loc, int b_glob, int nvalid_glob, float lambda_dssim, float lambda_flow,
                           float lambda_tv, const float* feat, float* g, float* loss_part, void* ws, hipStream_t st) {
    return unique_tensor_grad_impl(target, flows, masks, flow_shift, unq_inv, N, H, W, K, ids_unique, d_cidx, b_loc, b_glob, nvalid_glob, lambda_dssim,
                                   lambda_flow, lambda_tv, feat, g, loss_part, ws, nullptr, st);
}
static int unique_tensor_grad_impl(const float* target, const float* flows, const float* masks, const int* flow_shift, const int* unq_inv, int N, int H, int W, size_t K,
                                   int ids_unique, const int* d_cidx, int b_loc, int b_glob, int nvalid_glob, float lambda_dssim, float lambda_flow,
                                   float lambda_tv, const float* feat, float* g, float* loss_part, void* ws, const LazyGather* lz, hipStream_t st) {
    TCL_CHECK_ARG(target && flows && masks && flow_shift && unq_inv && d_cidx && feat && g && loss_part && ws);
    TCL_CHECK_ARG(N > 0 && b_loc > 0 && b_loc <= 64 && b_glob >= b_loc && nvalid_glob >= 0 && H > 160 && W > 160 && K > 0);
    StageWs S = carve_stage((char*)ws, b_loc, H, W);
    const size_t P = (size_t)H * W;
    const int b = b_loc;
    S.cidx = const_cast<int*>(d_cidx);
    if (hipMemsetAsync(S.acc, 0, ACC_SLOTS * 4 * sizeof(fx_t), st) != hipSuccess) return TCL_ELAUNCH;
    if (lz) hipLaunchKernelGGL(k_gather_codebook_lazy, pgrid(P, 2 * b), dim3(256), 0, st, feat, lz->m, lz->v, lz->t_last, unq_inv, S.cidx, S.cat, S.cmask, (int)P,
                               K, lz->upto, lz->lr, 0.9f, 0.999f, 1e-15f, lz->bc1, lz->bc2);
    else hipLaunchKernelGGL(k_gather_codebook, pgrid(P, 2 * b), dim3(256), 0, st, feat, unq_inv, S.cidx, S.cat, (int)P, K, S.cmask);
    const unsigned char* cmask = S.cmask;
    // loss = (1-lf)*ld*(1-msssim) + lf*flow + tv  -> fold (1-lf) into the ms-ssim lambda
    int rc = msssim_chain(S.cat, target, S.cidx, b * 3, H, W, (1.f - lambda_flow) * lambda_dssim, S.ms, true, st, b_glob * 3);
    if (rc) return rc;
    float ch = lambda_tv * 2.f / (3.f * (H - 1) * W) / b_glob, cw = lambda_tv * 2.f / (3.f * H * (W - 1)) / b_glob;
    hipLaunchKernelGGL(k_pixel_losses, pgrid(P, b * 3), dim3(256), 0, st, S.cat, (const float*)nullptr, S.cidx, S.ms.gX[1], S.ms.h[1],
                       S.ms.w[1], H, W, 0.f, ch, cw, S.gimg, S.acc);
    if (hipMemsetAsync(S.gpre, 0, (size_t)b * 3 * P * sizeof(fxq_t), st) != hipSuccess) return TCL_ELAUNCH;
    float inv_cnt = nvalid_glob ? 1.f / ((float)nvalid_glob * 3 * P) : 0.f;
    const float fscale = lambda_flow * inv_cnt;
    launch_flow_loss(S.cat, S.cidx, flows, masks, b, H, W, fscale, S.gimg, S.gpre, S.acc, pgrid(P, b), flow_shift, st);
    if (ids_unique)          // one cat row per launch, in order: conflict-free read-modify-write of the rows' gradients, no atomics
        for (int j = 0; j < 2 * b; ++j)
            hipLaunchKernelGGL(k_codebook_bwd<false>, pgrid(P, 1), dim3(256), 0, st, feat, unq_inv, S.cidx, S.gimg, S.gpre, fscale, flow_shift, b, j, g, (int)P, K, cmask);
    else hipLaunchKernelGGL(k_codebook_bwd<true>, pgrid(P, 2 * b), dim3(256), 0, st, feat, unq_inv, S.cidx, S.gimg, S.gpre, fscale, flow_shift, b, 0, g, (int)P, K, cmask);
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1), 0, st, S.acc, S.ms.term, 1.f, 0.f, lambda_flow, inv_cnt, ch, cw, loss_part);
    TCL_LAUNCH_RET();
}

// Stage 1 (generate.py:354-451).  sched: host int32 [iters][batch] frame ids, -1 pads a short batch;
// d_cat: packed cat indices on the device.  exposure/m/v/g: [N,3,4] device (exposure = eye, others 0 on entry).
// losses: device float [iters].  On return (stream order) `aligned_out` holds the aligned frames.
int tcl_exposure_align(const float* edited, const float* flows, const float* masks, const int* flow_shift, int N, int H, int W, const int* sched,
                       const int* d_cat, int iters, int iters_per_epoch, int batch, int epochs, float lr_init, float lr_final, float lambda_dssim,
                       float lambda_flow, float* exposure, float* g, float* m, float* v, float* losses, float* aligned_out,
                       void* ws, hipStream_t st) {
    TCL_CHECK_ARG(edited && flows && masks && flow_shift && sched && d_cat && exposure && g && m && v && losses && aligned_out && ws);
    TCL_CHECK_ARG(N > 0 && batch > 0 && iters > 0 && epochs > 0 && iters_per_epoch > 0 && H > 160 && W > 160);
    const size_t P = (size_t)H * W;
    const int total_iters = epochs * N / batch, per_epoch = iters_per_epoch;
    for (int it = 0; it < iters; ++it) {
        const int* bi = sched + (size_t)it * batch;
        int b = 0, nvalid = 0;
        while (b < batch && bi[b] >= 0) { nvalid += bi[b] > 0; ++b; }
        TCL_CHECK_ARG(b > 0);
        int epoch = it / per_epoch, i = it % per_epoch;
        float lr = (float)expon_lr(epoch * N / batch + i + 1, lr_init, lr_final, total_iters);
        int rc = tcl_exposure_grad(edited, flows, masks, flow_shift, N, H, W, d_cat + (size_t)it * 2 * batch, b, b, nvalid, lambda_dssim, lambda_flow,
                                   exposure, g, losses + it, ws, st);
        if (rc) return rc;
        rc = tcl_adam_step(exposure, g, m, v, (size_t)N * 12, lr, 0.9f, 0.999f, 1e-8f, it + 1, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_apply_exposure, pgrid(P, N), dim3(256), 0, st, edited, (const int*)nullptr, exposure, aligned_out, (int)P);
    TCL_LAUNCH_RET();
}

// Stage 2 (generate.py:453-533).  feat/g/m/v: channel-planar [3,K] device (feat initialised by tcl_scatter_mean_rgb2sh, others 0).
int tcl_unique_tensor_opt(const float* target, const float* flows, const float* masks, const int* flow_shift, const int* unq_inv, int N, int H, int W,
                          size_t K, int ids_unique, const int* sched, const int* d_cat, int iters, int batch, float feature_lr, float lambda_dssim, float lambda_flow,
                          float lambda_tv, float* feat, float* g, float* m, float* v, float* losses, float* images_out, void* ws,
                          void* lazy_ws, hipStream_t st) {
    TCL_CHECK_ARG(target && flows && masks && flow_shift && unq_inv && feat && g && m && v && losses && ws && (iters == 0 || (sched && d_cat)));
    TCL_CHECK_ARG(N > 0 && batch > 0 && batch <= 64 && iters >= 0 && H > 160 && W > 160 && K > 0);
    const size_t P = (size_t)H * W;
    const float lr = feature_lr * (float)batch / (float)N;
    // lazy dense Adam (see k_adam_catchup_frame): needs the frame-ordered row visits (ids_unique) and the caller's t_last / table scratch
    const bool lazy = lazy_ws && ids_unique && iters > 0;
    int* t_last = nullptr; float *bc1 = nullptr, *bc2 = nullptr;
    if (lazy) {
        t_last = (int*)lazy_ws;
        bc1 = (float*)((char*)lazy_ws + ((K * 4 + 255) & ~(size_t)255)); bc2 = bc1 + (iters + 1);
        std::vector<float> tab(2 * (size_t)(iters + 1), 1.f);                // bias corrections of steps 1 .. iters, as tcl_adam_step computes them
        for (int sidx = 1; sidx <= iters; ++sidx) { tab[sidx] = (float)(1.0 - pow((double)0.9f, sidx)); tab[iters + 1 + sidx] = (float)sqrt(1.0 - pow((double)0.999f, sidx)); }   // (double)b1 of the FLOAT b1, like tcl_adam_step
        if (hipMemsetAsync(t_last, 0, K * 4, st) != hipSuccess) return TCL_ELAUNCH;
        const size_t total = (size_t)N * P;
        TCL_CHECK_ARG(iters < TL_SHARED);
        if (total < 0x7fffffff) {
            hipLaunchKernelGGL(k_tl_mark, dim3(stream_grid((long)total, 256, 4)), dim3(256), 0, st, unq_inv, total, (int*)g);      // g is all zero here and is zeroed again below
            hipLaunchKernelGGL(k_tl_flag, dim3(stream_grid((long)total, 256, 4)), dim3(256), 0, st, unq_inv, total, (const int*)g, t_last);
            if (hipMemsetAsync(g, 0, K * 4, st) != hipSuccess) return TCL_ELAUNCH;
        } else if (hipMemsetD32Async((hipDeviceptr_t)t_last, TL_SHARED, K, st) != hipSuccess) return TCL_ELAUNCH;      // a clip of >= 2^31 pixels: the pixel numbers of the marking pass do not fit an int -- every row claimed with an atomic, as before
        if (upload_table(bc1, tab.data(), tab.size(), st) != TCL_OK) return TCL_ELAUNCH;
    }
    // run-ahead table (see k_adam_touched_frame): ahead[it][j] = the next iteration whose cat list ([cur | max(cur - 1, 0)], post_opt._pack_schedule) holds
    // the frame of cat row j of iteration `it`, or `iters`.  One backward sweep over the schedule.
    static const bool runahead = !(getenv("TCL_ADAM_RUNAHEAD") && atoi(getenv("TCL_ADAM_RUNAHEAD")) == 0);
    std::vector<int> ahead_tab;
    if (lazy) {
        ahead_tab.assign((size_t)iters * 2 * batch, 0);
        std::vector<int> nxt((size_t)N, iters);
        for (int it = iters - 1; it >= 0; --it) {
            const int* bi = sched + (size_t)it * batch;
            int b = 0;
            while (b < batch && bi[b] >= 0) ++b;
            for (int j = 0; j < 2 * b; ++j) {
                const int f = j < b ? bi[j] : (bi[j - b] > 0 ? bi[j - b] - 1 : 0);
                TCL_CHECK_ARG(f >= 0 && f < N);
                ahead_tab[(size_t)it * 2 * batch + j] = runahead ? nxt[f] : it + 1;
            }
            for (int j = 0; j < 2 * b; ++j) nxt[j < b ? bi[j] : (bi[j - b] > 0 ? bi[j - b] - 1 : 0)] = it;
        }
    }
    for (int it = 0; it < iters; ++it) {
        const int* bi = sched + (size_t)it * batch;
        int b = 0, nvalid = 0;
        while (b < batch && bi[b] >= 0) { nvalid += bi[b] > 0; ++b; }
        TCL_CHECK_ARG(b > 0);
        const int* cidx = d_cat + (size_t)it * 2 * batch;
        // lazy, default (rounds 3-5): a catch-up launch brings the mini-batch's rows to step `it` (writes them), the gather reads them, the step kernel
        // applies step it + 1.  TCL_ADAM_LAZY_V1=0 (round 5, opt-in): ONE visit with a write -- the gather replays the skipped steps in registers and the step
        // kernel replays them again before the gradient step.  Same bits; 208 -> 156 B of traffic per row, but the replay ARITHMETIC doubles: with the bench's
        // ~19 skipped steps per visit (one visit per epoch) the two kernels take 2.2 ms against 2.0 for catch-up + gather + step
        // (profiles/r5_bench_kernel_stats.txt vs r4), while on a 60-iteration run (short replays) it wins 5 % (profiles/r5_ab_path2_lazy_single_visit.txt).
        static const bool v1 = !(getenv("TCL_ADAM_LAZY_V1") && atoi(getenv("TCL_ADAM_LAZY_V1")) == 0);
        const LazyGather lz = {m, v, bc1, bc2, t_last, it, lr};
        if (lazy && v1)
            hipLaunchKernelGGL(k_adam_catchup_frame, pgrid(P, 2 * b), dim3(256), 0, st, unq_inv, cidx, (int)P, K, t_last, feat, m, v, it, lr, 0.9f, 0.999f,
                               1e-15f, bc1, bc2);
        int rc = unique_tensor_grad_impl(target, flows, masks, flow_shift, unq_inv, N, H, W, K, ids_unique, cidx, b, b, nvalid, lambda_dssim,
                                         lambda_flow, lambda_tv, feat, g, losses + it, ws, (lazy && !v1) ? &lz : nullptr, st);
        if (rc) return rc;
        if (lazy && !v1) {
            hipLaunchKernelGGL(k_adam_step_rows_lazy, pgrid(P, 2 * b), dim3(256), 0, st, unq_inv, cidx, (int)P, K, t_last, feat, g, m, v, it + 1, lr, 0.9f,
                               0.999f, 1e-15f, bc1, bc2);
        } else if (lazy) {
            const float c1 = (float)(1.0 - pow((double)0.9f, it + 1)), c2 = (float)sqrt(1.0 - pow((double)0.999f, it + 1));
            AheadArg ah;
            for (int j = 0; j < 2 * b; ++j) ah.v[j] = ahead_tab[(size_t)it * 2 * batch + j];
            hipLaunchKernelGGL(k_adam_touched_frame, pgrid(P, 2 * b), dim3(256), 0, st, unq_inv, cidx, (int)P, K, t_last, feat, g, m, v, it + 1, lr, 0.9f,
                               0.999f, 1e-15f, c1, c2, bc1, bc2, ah);
        } else {
            rc = tcl_adam_step(feat, g, m, v, K * 3, lr, 0.9f, 0.999f, 1e-15f, it + 1, st);
            if (rc) return rc;
        }
    }
    if (lazy) hipLaunchKernelGGL(k_adam_catchup_all, dim3(stream_grid((long)K, 256, 1)), dim3(256), 0, st, K, t_last, feat, m, v, iters, lr, 0.9f, 0.999f, 1e-15f, bc1, bc2);
    if (images_out) hipLaunchKernelGGL(k_gather_codebook, pgrid(P, N), dim3(256), 0, st, feat, unq_inv, (const int*)nullptr, images_out, (int)P, K, (unsigned char*)nullptr);
    TCL_LAUNCH_RET();
}

}  // extern "C"
