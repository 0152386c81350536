// fp16 MFMA GEMM for gfx950 with fused epilogues and an implicit-GEMM 3x3 convolution A-gather.
//   C[M,N] (f16) = act( A[M,K] (f16) . W[N,K]^T (f16) + bias[N] ) + residual[M,N]      (f32 accumulate)
// This single kernel family carries every dense contraction of path 1: Linear / conv1x1 (dense A),
// conv3x3 stride 1|2 with optional nearest-upsampled input (A rows gathered on the fly from NHWC
// activations, K = 9*Cin, weights tap-major, walked channel-slice-major: conv_kmap in gemm_conv.h), see SURVEY 8(a) A9.  Replaces the cuBLAS/cuDNN calls that
// diffusers' UNet2DConditionModel / AutoencoderKL make (reference call sites generate.py:342-347,
// utils/VidToMe/generate_utils.py:144,161).
//
// Tiling: BMxBNx64 block tile, 256 threads = WMxWN waves, each wave (BM/WM)x(BN/WN) out of 32x32x16 f16
// MFMAs; operands staged global -> VGPR -> LDS (16 B per lane, rows padded to 72 halves = 9 slots so the
// 16-lane ds_read_b128 groups are conflict-free).  ONE LDS buffer (36.8 KB) + register prefetch of the next K-step and
// __launch_bounds__(256, 3): measured on MI355X, 3 resident blocks per CU hide latency better than a second LDS buffer or a
// deeper register pipeline (PF = 2/3 drop to 2 waves/SIMD and lose 10-40 %).  The epilogue stages f16 tiles through LDS so
// stores (and residual loads) are whole 16-B row chunks.  Small-M / deep-K problems (1280-channel levels, yt-plane chunks)
// are split along K into f32 partials with a deterministic second pass.
// Blocks are laid out XCD-aware: XCD x owns M-tiles == x (mod 8) and walks N-tiles fastest so the A tile
// stays in that XCD's L2 while W (small) is L2-resident everywhere.
#include "common.h"
#include <stdio.h>
#include <string.h>
#include "../../include/tclight_hip.h"
#include <stdlib.h>
#include <math.h>
#include <unordered_map>
#include "gemm_conv.h"
#include "prof.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define BK 64
#define LDS_STRIDE 72  // halves


template <int BM, int BN, int WM, int WN, int PF>
__global__ __launch_bounds__(256, PF == 1 ? 3 : 2) void k_gemm(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                              const _Float16* __restrict__ bias, const _Float16* __restrict__ resid,
                                              _Float16* __restrict__ C, int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act,
                                              ConvP cp, int tiles_m, int tiles_n, int nk_per, float* __restrict__ part) {
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    constexpr int A_IT = BM * 8 / 256, B_IT = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* As = (_Float16*)smem;                          // [BM][72]   (single LDS buffer, register prefetch)
    _Float16* Bs = As + BM * LDS_STRIDE;                     // [BN][72]

    // XCD-aware tile assignment
    const int tgrid = ((tiles_m + 7) >> 3) * 8 * tiles_n;        // blocks per K-split
    const int split = blockIdx.x / tgrid;
    const int bid = blockIdx.x - split * tgrid, xcd = bid & 7, j = bid >> 3;
    const int tn = j % tiles_n, tm = (j / tiles_n) * 8 + xcd;
    if (tm >= tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;

    // per-thread global-load descriptors: chunk c = tid + 256*i -> row c/8, k-chunk c%8 (8 halves)
    const int kc8 = (tid & 7) * 8;
    long a_off[A_IT]; int a_oy[A_IT], a_ox[A_IT]; bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + (tid >> 3) + 32 * i;
        a_ok[i] = m < M;
        if (!cp.conv) { a_off[i] = (long)m * lda; a_oy[i] = a_ox[i] = 0; }
        else {
            int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
            a_off[i] = (long)b * cp.Hin * cp.Win * cp.Cin; a_oy[i] = oy * cp.stride - cp.pad; a_ox[i] = ox * cp.stride - cp.pad;
        }
    }
    const _Float16* wp[B_IT]; bool w_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) { int n = n0 + (tid >> 3) + 32 * i; w_ok[i] = n < N; wp[i] = W + (long)(w_ok[i] ? n : 0) * ldw + kc8; }

    // PF register staging sets (native vector type: hipcc keeps HIP's uint4 struct arrays in scratch): tile t+1..t+PF in
    // flight while tile t (already in the single LDS buffer) is consumed.
    u32x4 ra[PF][A_IT], rb[PF][B_IT];
#define GEMM_GLOAD(KT, S)                                                                                                     \
    {                                                                                                                         \
        const int k0_ = (KT) * BK;                                                                                            \
        int tdy_ = 0, tdx_ = 0, c0_ = k0_, kw_ = k0_;                                                                         \
        if (cp.conv) { int tap_ = conv_kmap(k0_, cp.Cin, c0_); tdy_ = tap_ / 3; tdx_ = tap_ - tdy_ * 3; kw_ = tap_ * cp.Cin + c0_; } \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                                   \
            u32x4 v_ = {0u, 0u, 0u, 0u};                                                                                      \
            if (a_ok[i]) {                                                                                                    \
                if (!cp.conv) v_ = *(const u32x4*)(A + a_off[i] + k0_ + kc8);                                                 \
                else {                                                                                                        \
                    int iy_ = a_oy[i] + tdy_, ix_ = a_ox[i] + tdx_;                                                           \
                    if (iy_ >= 0 && iy_ < cp.Hup && ix_ >= 0 && ix_ < cp.Wup) {                                               \
                        if (cp.Hup != cp.Hin || cp.Wup != cp.Win) { iy_ = min((int)floorf(iy_ * cp.sy), cp.Hin - 1); ix_ = min((int)floorf(ix_ * cp.sx), cp.Win - 1); } \
                        v_ = *(const u32x4*)(A + a_off[i] + ((long)iy_ * cp.Win + ix_) * cp.Cin + c0_ + kc8);                 \
                    }                                                                                                         \
                }                                                                                                             \
            }                                                                                                                 \
            ra[S][i] = v_;                                                                                                    \
        }                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) { u32x4 z_ = {0u, 0u, 0u, 0u}; rb[S][i] = w_ok[i] ? *(const u32x4*)(wp[i] + kw_) : z_; } \
    }
#define GEMM_SSTORE(S)                                                                                                        \
    {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) *(u32x4*)(As + ((tid >> 3) + 32 * i) * LDS_STRIDE + kc8) = ra[S][i]; \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) *(u32x4*)(Bs + ((tid >> 3) + 32 * i) * LDS_STRIDE + kc8) = rb[S][i]; \
    }

    float16v acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int kt0 = split * nk_per, nk = min(K / BK, kt0 + nk_per);
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (kt0 + u < nk) GEMM_GLOAD(kt0 + u, u);
    GEMM_SSTORE(0);
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const _Float16* as = As + (wm * (BM / WM) + frow) * LDS_STRIDE + fk;
    const _Float16* bs = Bs + (wn * (BN / WN) + frow) * LDS_STRIDE + fk;
    for (int kt = kt0; kt < nk; kt += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = kt + u;                    // tile t is in LDS; set u is free, sets u+1.. hold tiles t+1..
            if (t < nk) {
                if (t + PF < nk) GEMM_GLOAD(t + PF, u);
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    half8 fa[MT], fb[NT];
#pragma unroll
                    for (int a = 0; a < MT; ++a) fa[a] = *(const half8*)(as + a * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
                    for (int b = 0; b < NT; ++b) fb[b] = *(const half8*)(bs + b * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
                }
                __syncthreads();                     // everyone done reading the buffer before it is overwritten
                if (t + 1 < nk) GEMM_SSTORE((u + 1) % PF);
                __syncthreads();
            }
        }
    }
#undef GEMM_GLOAD
#undef GEMM_SSTORE

    // ---- epilogue A (vector path): stage f16(act(acc+bias)) through LDS, then whole 16-B row chunks (+residual) to HBM
    constexpr int CS = BN + 8;                      // staging row stride (halves); BM*CS*2 bytes <= the operand buffers
    if (!part && (N & 7) == 0 && (ldc & 7) == 0 && (!resid || (ldr & 7) == 0)) {
        _Float16* Cs = (_Float16*)smem;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int nl = wn * (BN / WN) + b * 32 + (lane & 31), n = n0 + nl;
            const float bv = (bias && n < N) ? (float)bias[n] : 0.f;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int ml = wm * (BM / WM) + a * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[a][b][r] + bv;
                    v = apply_act(v, act);
                    Cs[(ml + (r & 3) + 8 * (r >> 2)) * CS + nl] = (_Float16)v;
                }
            }
        }
        __syncthreads();
        if (act == 2) {     // GEGLU (diffusers GEGLU, ff.net.0): every 64-column group holds [32 value | 32 gate] of the same output columns
            constexpr int CPH = BN / 16;
#pragma unroll
            for (int i = 0; i < BM * CPH / 256; ++i) {
                const int c = tid + 256 * i, row = c / CPH, c8 = (c % CPH) * 8, m = m0 + row, n = (n0 >> 1) + c8;
                if (m < M && n < (N >> 1)) {
                    const int gc = (c8 >> 5) * 64 + (c8 & 31);           // 64-column groups [32 value | 32 gate]
                    half8 va = *(const half8*)(Cs + row * CS + gc), vg = *(const half8*)(Cs + row * CS + gc + 32);
#pragma unroll
                    for (int q = 0; q < 8; ++q) { float gf = (float)vg[q]; va[q] = (_Float16)((float)va[q] * gelu_erf(gf)); }
                    *(half8*)(C + (long)m * ldc + n) = va;
                }
            }
            return;
        }
        constexpr int CPR = BN / 8;
#pragma unroll
        for (int i = 0; i < BM * CPR / 256; ++i) {
            const int c = tid + 256 * i, row = c / CPR, c8 = (c % CPR) * 8, m = m0 + row, n = n0 + c8;
            if (m < M && n < N) {
                half8 v = *(const half8*)(Cs + row * CS + c8);
                if (resid) {
                    half8 rv = *(const half8*)(resid + (long)m * ldr + n);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (_Float16)post_act((float)v[q] + (float)rv[q], act);
                }
                *(half8*)(C + (long)m * ldc + n) = v;
            }
        }
        return;
    }
    // ---- epilogue B (scalar path; split-K partials, odd N): lane holds column n = lane&31 and rows (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + wn * (BN / WN) + b * 32 + (lane & 31);
        if (n >= N) continue;
        const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int mb = m0 + wm * (BM / WM) + a * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m >= M) continue;
                if (part) { part[((long)split * M + m) * N + n] = acc[a][b][r]; continue; }   // split-K partial (f32)
                float v = acc[a][b][r] + bv;
                v = apply_act(v, act);
                if (resid) v = post_act((float)(_Float16)v + (float)resid[(long)m * ldr + n], act);
                C[(long)m * ldc + n] = (_Float16)v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (128x128x32 steps): operands go global -> LDS directly (global_load_lds_dwordx4: the 64 lanes of a wave
// write 1 KiB contiguous = 16 rows x 64 B), no VGPR staging and no ds_write pass, two 16 KiB stages (one barrier per K-step).
// Rows are unpadded, so the 16-B chunk index of row r is XOR-swizzled with (r>>2)&3 -- applied to the per-lane SOURCE address
// and to the ds_read address (cdna guide rule 21) -- which makes the 16-lane ds_read_b128 groups conflict-free.
// Out-of-range rows / conv taps fetch from a zero page.  ~110 VGPRs and 34.8 KiB LDS -> 4 blocks per CU.
__device__ __attribute__((aligned(16))) unsigned g_zero_page[64];

// QP (round 5): the per-chunk QKV projection of a merging transformer block with its result written straight into the attention panels (QkvPanel,
// gemm_conv.h) -- M tiles are cut per batch entry (tile rows = 128 consecutive tokens of ONE entry = two whole 64-key V^T tiles), the epilogue sends
// the Q / K row chunks to their head-major rows and transposes the V columns through the staged C tile; k_pack_qkv and the [M, 3 C] round trip go.
template <int BM, int BN, int WM, int WN, int STAGES, int QP = 0>
__global__ __launch_bounds__(256, BM * BN >= 256 * 256 ? 1 : BM * BN > 128 * 128 ? 2 : BM * BN < 128 * 128 ? 4 : (STAGES == 2 ? 4 : 3)) void k_gemm_dma(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                     const _Float16* __restrict__ bias, const _Float16* __restrict__ resid,
                                                     _Float16* __restrict__ C, int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act,
                                                     ConvP cp, int tiles_m, int tiles_n, int nk_per, float* __restrict__ part, int xcd_n, QkvPanel qp) {
    constexpr int KB = 32;                                  // K per step
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    constexpr int ROWB = KB * 2;                            // bytes per LDS row (64)
    constexpr int STAGE = (BM + BN) * ROWB;                 // 16 KiB
    constexpr int A_IT = BM / 16 / 4, B_IT = BN / 16 / 4;   // 1-KiB pieces (16 rows) per wave per operand
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order (block id % 8 = XCD): normally XCD x owns the M-tiles == x (mod 8) and walks N fastest, so an A panel
    // stays in one L2 and W (small) is resident in all of them; xcd_n swaps the roles for weight-dominated problems (N > M), where
    // each W panel should be fetched from HBM by one XCD only.
    int tm, tn, split;
    if (!xcd_n) {
        const int tgrid = ((tiles_m + 7) >> 3) * 8 * tiles_n;        // blocks per K-split
        split = blockIdx.x / tgrid;
        const int bid = blockIdx.x - split * tgrid, xcd = bid & 7, j = bid >> 3;
        tn = j % tiles_n; tm = (j / tiles_n) * 8 + xcd;
    } else {
        const int tgrid = ((tiles_n + 7) >> 3) * 8 * tiles_m;
        split = blockIdx.x / tgrid;
        const int bid = blockIdx.x - split * tgrid, xcd = bid & 7, j = bid >> 3;
        tm = j % tiles_m; tn = (j / tiles_m) * 8 + xcd;
    }
    if (tm >= tiles_m || tn >= tiles_n) return;
    int m0 = tm * BM, mlim = M, qb = 0, qtl = 0;
    if constexpr (QP) {                       // tile tm = (entry qb, token tile qtl): rows [qb T + qtl BM, ...) of that entry only
        const int tpe = (qp.T + BM - 1) / BM;
        qb = tm / tpe; qtl = tm - qb * tpe;
        m0 = qb * qp.T + qtl * BM; mlim = (qb + 1) * qp.T;
    }
    const int n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;

    // DMA lane roles: piece p (16 rows) of an operand; lane -> row rr = lane>>2, LDS chunk c' = lane&3, source chunk c = c' ^ ((rr>>2)&3)
    const int rr = lane >> 2, csrc = ((lane & 3) ^ ((rr >> 2) & 3)) * 8;          // source offset in halves within the 32-wide K slice
    const _Float16* zero = (const _Float16*)g_zero_page;
    long a_off[A_IT]; int a_oy[A_IT], a_ox[A_IT]; bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + (wid * A_IT + i) * 16 + rr;
        a_ok[i] = m < mlim;
        if constexpr (QP) {                                   // token t of entry qb: row aidx[t] (or t) of the entry's source block
            const int t = qtl * BM + (wid * A_IT + i) * 16 + rr;
            a_off[i] = (long)qb * qp.a_bs + (long)(qp.aidx && a_ok[i] ? qp.aidx[t] : t) * lda; a_oy[i] = a_ox[i] = 0;
        } else
        if (!cp.conv) { a_off[i] = (long)m * lda; a_oy[i] = a_ox[i] = 0; }
        else {
            int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
            a_off[i] = (long)b * cp.Hin * cp.Win * cp.Cin; a_oy[i] = oy * cp.stride - cp.pad; a_ox[i] = ox * cp.stride - cp.pad;
        }
    }
    const _Float16* wp[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) { int n = n0 + (wid * B_IT + i) * 16 + rr; wp[i] = n < N ? W + (long)n * ldw + csrc : nullptr; }

#define DMA_ISSUE(KT, BUF)                                                                                                    \
    {                                                                                                                         \
        const int k0_ = (kbeg + (KT)) * KB;                                                                                   \
        int tdy_ = 0, tdx_ = 0, c0_ = k0_, kw_ = k0_;                                                                         \
        if (cp.conv) { int tap_ = conv_kmap(k0_, cp.Cin, c0_); tdy_ = tap_ / 3; tdx_ = tap_ - tdy_ * 3; kw_ = tap_ * cp.Cin + c0_; } \
        char* sb_ = smem + (BUF) * STAGE;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                                   \
            const _Float16* src_ = zero;                                                                                      \
            if (a_ok[i]) {                                                                                                    \
                if (!cp.conv) src_ = A + a_off[i] + k0_ + csrc;                                                               \
                else {                                                                                                        \
                    int iy_ = a_oy[i] + tdy_, ix_ = a_ox[i] + tdx_;                                                           \
                    if (iy_ >= 0 && iy_ < cp.Hup && ix_ >= 0 && ix_ < cp.Wup) {                                               \
                        if (cp.Hup != cp.Hin || cp.Wup != cp.Win) { iy_ = min((int)floorf(iy_ * cp.sy), cp.Hin - 1); ix_ = min((int)floorf(ix_ * cp.sx), cp.Win - 1); } \
                        src_ = A + a_off[i] + ((long)iy_ * cp.Win + ix_) * cp.Cin + c0_ + csrc;                               \
                    }                                                                                                         \
                }                                                                                                             \
            }                                                                                                                 \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                             \
                                             (__attribute__((address_space(3))) void*)(sb_ + (wid * A_IT + i) * 1024), 16, 0, 0); \
        }                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                                                   \
            const _Float16* src_ = wp[i] ? wp[i] + kw_ : zero;                                                                \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                             \
                                             (__attribute__((address_space(3))) void*)(sb_ + BM * ROWB + (wid * B_IT + i) * 1024), 16, 0, 0); \
        }                                                                                                                     \
    }

    float16v acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int kbeg = split * nk_per, nk = min(K / KB, kbeg + nk_per) - kbeg;      // this block's K steps: [kbeg, kbeg + nk)
    const int frow = lane & 31, fh = lane >> 5;
    // fragment read: row R = tile row (lane&31), logical chunk c = 2*ks + (lane>>5), physical chunk c ^ ((R>>2)&3)
#define DMA_COMPUTE(BUF)                                                                                                      \
    {                                                                                                                         \
        const char* ab = smem + (BUF) * STAGE + (wm * (BM / WM)) * ROWB;                                                      \
        const char* bb = smem + (BUF) * STAGE + BM * ROWB + (wn * (BN / WN)) * ROWB;                                          \
        _Pragma("unroll") for (int ks = 0; ks < KB / 16; ++ks) {                                                             \
            half8 fa[MT], fb[NT];                                                                                             \
            _Pragma("unroll") for (int a = 0; a < MT; ++a) { int R = a * 32 + frow; fa[a] = *(const half8*)(ab + R * ROWB + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); } \
            _Pragma("unroll") for (int b = 0; b < NT; ++b) { int R = b * 32 + frow; fb[b] = *(const half8*)(bb + R * ROWB + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); } \
            _Pragma("unroll") for (int a = 0; a < MT; ++a)                                                                    \
                _Pragma("unroll") for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0); \
        }                                                                                                                     \
    }
    if constexpr (STAGES == 2) {
        DMA_ISSUE(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) DMA_ISSUE(kt + 1, cur ^ 1);
            DMA_COMPUTE(cur);
            __syncthreads();
        }
    } else {
        // 3-stage ring, prefetch distance 2, ONE raw barrier per step: wait for my own pieces of tile t (counted vmcnt: the newer
        // tile t+1 stays in flight), barrier (=> every wave's pieces landed AND everyone left tile t-1), refill the stage tile t-1
        // used with tile t+2, compute tile t.  Raw s_barrier: __syncthreads() would drain the DMA queue (vmcnt(0)).
        constexpr int PER_TILE = A_IT + B_IT;            // glds instructions per wave per tile
        DMA_ISSUE(0, 0);
        if (nk > 1) DMA_ISSUE(1, 1);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) { const int nb = buf == 0 ? 2 : buf - 1; DMA_ISSUE(kt + 2, nb); }
            DMA_COMPUTE(buf);
            buf = buf == 2 ? 0 : buf + 1;
        }
        __syncthreads();                                  // all reads done before the epilogue reuses the LDS
    }
#undef DMA_COMPUTE
#undef DMA_ISSUE
    if (part) {     // split-K partial sums (f32): lane holds column n = lane&31 and rows (r&3)+8*(r>>2)+4*(lane>>5) of each 32x32 tile
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = n0 + wn * (BN / WN) + b * 32 + (lane & 31);
            if (n >= N) continue;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int mb = m0 + wm * (BM / WM) + a * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < M) part[((long)split * M + m) * N + n] = acc[a][b][r];
                }
            }
        }
        return;
    }
    // epilogue: same LDS-staged vector path as k_gemm (callers guarantee N % 8 == 0 etc. before choosing this kernel)
    constexpr int CS = BN + 8;
    _Float16* Cs = (_Float16*)smem;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int nl = wn * (BN / WN) + b * 32 + (lane & 31), n = n0 + nl;
        const float bv = (bias && n < N) ? (float)bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int ml = wm * (BM / WM) + a * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][b][r] + bv;
                v = apply_act(v, act);
                Cs[(ml + (r & 3) + 8 * (r >> 2)) * CS + nl] = (_Float16)v;
            }
        }
    }
    __syncthreads();
    if constexpr (QP) {
        static_assert(!QP || BM == 128, "two whole 64-key tiles per M tile");
        const int Cc = qp.H * qp.d, t0 = qtl * BM;
        constexpr int CPRQ = BN / 8;
#pragma unroll
        for (int i = 0; i < BM * CPRQ / 256; ++i) {                 // Q and K: 8-column row chunks (d % 8 == 0: a chunk never straddles a head or a projection)
            const int c = tid + 256 * i, row = c / CPRQ, c8 = (c % CPRQ) * 8, t = t0 + row, n = n0 + c8;
            if (t >= qp.T || n >= 2 * Cc) continue;
            half8 v = *(const half8*)(Cs + row * CS + c8);
            const int which = n >= Cc, nn = n - which * Cc, head = nn / qp.d, dd = nn - head * qp.d;
            const long bh = (long)qb * qp.H + head;
            if (!which) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (_Float16)((float)v[q] * qp.qscale);
                *(half8*)(qp.qp + (bh * qp.Tqp + t) * qp.DP + dd) = v;
            } else {
                _Float16* kr = qp.kp + (bh * qp.Tkp + t) * qp.KS;
                *(half8*)(kr + dd) = v;
                if (qp.one_col >= 0 && dd + 8 == qp.d) { half8 o1; o1[0] = (_Float16)1.f;
#pragma unroll
                    for (int q = 1; q < 8; ++q) o1[q] = (_Float16)0.f;
                    *(half8*)(kr + qp.one_col) = o1; }
            }
        }
        const int v_lo = max(n0, 2 * Cc), v_hi = min(n0 + BN, N);   // the V columns of this tile -> V^T tile rows, 8 permuted key positions per store
        if (v_lo < v_hi) {
            const int nt = qp.Tkp / 64, items = (v_hi - v_lo) * 16;
            for (int it = tid; it < items; it += 256) {
                const int col = it >> 4, kq = (it >> 3) & 1, p8 = it & 7;
                const int kt = qtl * (BM / 64) + kq;
                if (kt >= nt) continue;
                const int n = v_lo + col, nn = n - 2 * Cc, head = nn / qp.d, dd = nn - head * qp.d;
                _Float16* tile = qp.vt + (((long)qb * qp.H + head) * nt + kt) * qp.vtile;
                const _Float16* src = Cs + (kq * 64) * CS + (n - n0);
                const bool ones_too = dd + 1 == qp.d && qp.DPV > qp.d;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    if (pass == 1 && !ones_too) break;
                    const int rowi = pass ? qp.d : dd;
                    const int sk = (qp.skew && (((rowi & 15) + 4) & 8)) ? 16 : 0;
                    half8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ps = (8 * p8 + j) ^ sk, r = (ps & ~12) | ((ps & 4) << 1) | ((ps & 8) >> 1);
                        const bool ok = kt * 64 + r < qp.T;
                        o[j] = ok ? (pass ? (_Float16)1.f : src[r * CS]) : (_Float16)0.f;
                    }
                    *(half8*)(tile + rowi * 72 + 8 * p8) = o;
                }
            }
        }
        return;
    }
    if (act == 2) {
        constexpr int CPH = BN / 16;
#pragma unroll
        for (int i = 0; i < BM * CPH / 256; ++i) {
            const int c = tid + 256 * i, row = c / CPH, c8 = (c % CPH) * 8, m = m0 + row, n = (n0 >> 1) + c8;
            if (m < M && n < (N >> 1)) {
                const int gc = (c8 >> 5) * 64 + (c8 & 31);               // 64-column groups [32 value | 32 gate]
                half8 va = *(const half8*)(Cs + row * CS + gc), vg = *(const half8*)(Cs + row * CS + gc + 32);
#pragma unroll
                for (int q = 0; q < 8; ++q) { float gf = (float)vg[q]; va[q] = (_Float16)((float)va[q] * gelu_erf(gf)); }
                *(half8*)(C + (long)m * ldc + n) = va;
            }
        }
        return;
    }
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int i = 0; i < BM * CPR / 256; ++i) {
        const int c = tid + 256 * i, row = c / CPR, c8 = (c % CPR) * 8, m = m0 + row, n = n0 + c8;
        if (m < M && n < N) {
            half8 v = *(const half8*)(Cs + row * CS + c8);
            if (resid) {
                half8 rv = *(const half8*)(resid + (long)m * ldr + n);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (_Float16)post_act((float)v[q] + (float)rv[q], act);
            }
            *(half8*)(C + (long)m * ldc + n) = v;
        }
    }
}

// caller-owned scratch for split-K partials (tcl_set_workspace); all GEMMs using it must be issued on one stream
static float* g_ws = nullptr;
static size_t g_ws_bytes = 0;
__global__ void k_splitk_finalize(const float* __restrict__ part, int splits, const _Float16* __restrict__ bias, const _Float16* __restrict__ resid,
                                  _Float16* __restrict__ C, int M, int N, int ldc, int ldr, int act);

template <int BM, int BN, int WM, int WN, int STAGES>
static int launch_gemm_dma(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N,
                           int K, int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st, int splits = 1) {
    const int tm = cdiv(M, BM), tn = cdiv(N, BN), nk = K / 32;
    const size_t ops = (size_t)STAGES * (BM + BN) * 64, cs = (size_t)BM * (BN + 8) * 2, lds = ops > cs ? ops : cs;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_gemm_dma<BM, BN, WM, WN, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    if (splits > nk / 4) splits = nk / 4 > 0 ? nk / 4 : 1;
    while (splits > 1 && (!g_ws || (size_t)splits * M * N * 4 > g_ws_bytes)) --splits;
    if (act == 2) splits = 1;
    const int nk_per = cdiv(nk, splits);
    splits = cdiv(nk, nk_per);
    float* part = splits > 1 ? g_ws : nullptr;
    static const int xcd_mode = getenv("TCL_GEMM_XCDN") ? atoi(getenv("TCL_GEMM_XCDN")) : -1;      // -1 auto, 0/1 forced (experiments)
    const int xcd_n = xcd_mode < 0 ? (N > M) : xcd_mode;
    const int grid = (xcd_n ? cdiv(tn, 8) * 8 * tm : cdiv(tm, 8) * 8 * tn) * splits;
    hipLaunchKernelGGL((k_gemm_dma<BM, BN, WM, WN, STAGES>), dim3(grid), dim3(256), lds, st, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr,
                       act, cp, tm, tn, nk_per, part, xcd_n, QkvPanel{});
    if (part) hipLaunchKernelGGL(k_splitk_finalize, dim3(stream_grid((long)M * N, 256, 4)), dim3(256), 0, st, part, splits, bias, resid, C, M, N, ldc, ldr, act);
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

// the QKV projection of ne x T merged tokens into the attention panels: 128 x 128 tiles, M tiles per batch entry, no split-K
int gemm_dma_qkv_panels(const _Float16* A, const _Float16* W, int ne, int K, int lda, int ldw, const QkvPanel& qp, hipStream_t st) {
    constexpr int BM = 128, BN = 128, STAGES = 3;
    const int N = 3 * qp.H * qp.d, M = ne * qp.T;
    if (K % 32 != 0 || (lda & 7) || (ldw & 7) || qp.d % 8 != 0 || qp.Tkp % 64 != 0) return TCL_EINVAL;
    const int tm = ne * cdiv(qp.T, BM), tn = cdiv(N, BN);
    const size_t ops = (size_t)STAGES * (BM + BN) * 64, cs = (size_t)BM * (BN + 8) * 2, lds = ops > cs ? ops : cs;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_gemm_dma<BM, BN, 2, 2, STAGES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    ConvP cp = {};
    const int grid = cdiv(tm, 8) * 8 * tn;
    hipLaunchKernelGGL((k_gemm_dma<BM, BN, 2, 2, STAGES, 1>), dim3(grid), dim3(256), lds, st, A, W, (const _Float16*)nullptr, (const _Float16*)nullptr,
                       (_Float16*)nullptr, M, N, K, lda, ldw, N, N, 0, cp, tm, tn, K / 32, (float*)nullptr, 0, qp);
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

// split-K second pass: C = act(sum_s part[s] + bias) + resid
__global__ void k_splitk_finalize(const float* __restrict__ part, int splits, const _Float16* __restrict__ bias, const _Float16* __restrict__ resid,
                                  _Float16* __restrict__ C, int M, int N, int ldc, int ldr, int act) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N); const long m = i / N;
        float v = bias ? (float)bias[n] : 0.f;
        for (int sidx = 0; sidx < splits; ++sidx) v += part[(long)sidx * total + i];
        v = apply_act(v, act);
        if (resid) v = post_act(v + (float)resid[m * ldr + n], act);
        C[m * ldc + n] = (_Float16)v;
    }
}

template <int BM, int BN, int WM, int WN, int PF>
static int launch_gemm(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N,
                       int K, int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st) {
    const int tm = cdiv(M, BM), tn = cdiv(N, BN);
    const int grid = cdiv(tm, 8) * 8 * tn;
    const size_t lds = (size_t)(BM + BN) * LDS_STRIDE * 2;
    static_assert((size_t)BM * (BN + 8) * 2 <= (size_t)(BM + BN) * LDS_STRIDE * 2, "C staging must fit");
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_gemm<BM, BN, WM, WN, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    // small-M / deep-K problems (the 1280-channel levels, yt-plane chunks) leave most of the 256 CUs idle: split K so that
    // ~2 blocks per CU exist, partials in f32, deterministic second pass.
    const int nk = K / BK;
    int splits = 1;
    const int sk_tiles = 700, sk_target = 1024;
    if (tm * tn < sk_tiles && nk >= 32 && g_ws && act != 2) {     // measured: splitting K < 2048 loses to the extra pass
        splits = min(nk / 8, cdiv(sk_target, tm * tn));
        while (splits > 1 && (size_t)splits * M * N * 4 > g_ws_bytes) --splits;
    }
    const int nk_per = cdiv(nk, splits);
    splits = cdiv(nk, nk_per);
    float* part = splits > 1 ? g_ws : nullptr;
    hipLaunchKernelGGL((k_gemm<BM, BN, WM, WN, PF>), dim3(grid * splits), dim3(256), lds, st, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp,
                       tm, tn, nk_per, part);
    if (part) hipLaunchKernelGGL(k_splitk_finalize, dim3(stream_grid((long)M * N, 256, 4)), dim3(256), 0, st, part, splits, bias, resid, C, M, N, ldc, ldr, act);
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

// ---- configuration choice.  cfg ids (also the tcl_gemm_tune ids):
//   1 dma 128x128   2 dma 64x128   3 dma 128x64   4 dma 64x64   11 dma 256x128      (k_gemm_dma, 3 stages, optional split-K)
//   5 g8 256x320    6 g8 128x320   7 g8 256x256   8 g8 128x256                      (gemm8.hip, 8-wave ping-pong)
//   9 reg 128x128   10 reg 128x64                                                    (k_gemm, register-staged; any N / ld)
//   12 strip-resident K = 320 Linear (linstrip.hip: 128 activation rows in registers, weight rows swept through LDS)
//   13 q8 256x256   14 q8 256x320   15 q8 512x128                                                    (gemm8q.hip, 8-phase 256-row kernels, round 4)
static int g_tune_cfg = 0, g_tune_splits = 0;

static int run_cfg(int cfg, int splits, const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N,
                   int K, int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st) {
    switch (cfg) {
        case 1: return launch_gemm_dma<128, 128, 2, 2, 3>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st, splits);
        case 2: return launch_gemm_dma<64, 128, 2, 2, 3>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st, splits);
        case 3: return launch_gemm_dma<128, 64, 2, 2, 3>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st, splits);
        case 4: return launch_gemm_dma<64, 64, 2, 2, 3>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st, splits);
        case 11: return launch_gemm_dma<256, 128, 2, 2, 3>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st, splits);
        case 5: case 6: case 7: case 8: return gemm8_dispatch(cfg - 4, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
        case 9: return launch_gemm<128, 128, 2, 2, 1>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
        case 10: return launch_gemm<128, 64, 4, 1, 1>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
        case 13: case 14: case 15: return gemm8q_dispatch(cfg - 12, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
        case 12:
            // the strip kernel's last weight tile is moved back when N % 128 != 0 and re-reads the residual of the overlapped columns: with an
            // in-place residual (resid == C) those columns would get it twice -> such a call takes the tiled kernel
            if (resid == C && N % 128 != 0) return run_cfg((N % 128 == 0 || N > 192) ? 1 : 3, splits, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
            return lin_strip_dispatch(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, st);
    }
    return TCL_EINVAL;
}

static bool cfg_ok(int cfg, int M, int N, int K, int ldc, int ldr, bool has_resid, int act, const ConvP& cp) {
    const bool vec_ok = (N & 7) == 0 && (ldc & 7) == 0 && (!has_resid || (ldr & 7) == 0) && (K % 32) == 0;
    if (cfg == 9 || cfg == 10) return act != 2 || cfg == 9;
    if (!vec_ok) return false;
    if (cfg == 12) return lin_strip_ok(M, N, K, 8, 8, ldc, ldr, has_resid, act, cp);     // lda / ldw: multiples of 8 by tcl_gemm_f16's argument check
    if (cfg >= 13 && cfg <= 15) return gemm8q_ok(cfg - 12, M, N, K, 8, 8, ldc, ldr, has_resid, act, cp);
    if (cfg >= 5 && cfg <= 8) return (act != 2 || cfg >= 7) && K % 64 == 0 && (!cp.conv || cp.Cin % 64 == 0);   // GEGLU: 64-column wave strips only
    if (act == 2) return true;                                          // GEGLU epilogue: 64-column [value | gate] groups, every BN is a multiple
    return true;
}

// Which tiles the tuner may time for a shape -- and which a persistent table entry may name (a stale or hand-edited table must
// not route a shape to a tile that does not divide N or that drops the K split its candidates were measured with).
static bool tile_ok(int cfg, int M, int N, int K, int splits) {
    const int t128 = cdiv(M, 128) * cdiv(N, 128);
    switch (cfg) {
        case 1: return true;
        case 2: case 3: return t128 < 4096;
        case 4: return t128 < 1024;
        case 11: return t128 >= 256;
        case 5: return splits == 1 && K >= 512 && N % 320 == 0 && cdiv(M, 256) * (N / 320) >= 96;
        case 6: return splits == 1 && K >= 512 && N % 320 == 0 && cdiv(M, 128) * (N / 320) >= 96;
        case 7: return splits == 1 && K >= 512 && N % 256 == 0 && cdiv(M, 256) * (N / 256) >= 96;
        case 8: return splits == 1 && K >= 512 && N % 256 == 0 && cdiv(M, 128) * (N / 256) >= 96;
        case 12: return splits == 1 && K == 320 && N >= 128 && M >= 16384;
        case 13: return splits == 1 && K >= 320 && N % 256 == 0 && cdiv(M, 256) * (N / 256) >= 96;
        case 14: return splits == 1 && K >= 320 && N % 320 == 0 && cdiv(M, 256) * (N / 320) >= 96;
        case 15: return splits == 1 && K >= 320 && N % 128 == 0 && N % 256 != 0 && cdiv(M, 512) * (N / 128) >= 96;
    }
    return false;
}

// ---- automatic choice: a per-process cache keyed by the problem shape, filled on first use by timing the valid candidates on the
// caller's stream (hipEvents; the tuning call synchronises the stream, later calls are a hash lookup).  Numerics do not depend
// on the outcome: every tile configuration accumulates each output element over k in the same order (16-wide MFMA blocks in
// sequence, f32), and the number of K splits -- the only thing that changes the summation order -- is a fixed function of
// the shape (split_rule), identical for all candidates.
struct TuneKey {
    int conv, M, N, K, act, hasr, Hin, Win, Cin, stride, Hup;
    bool operator==(const TuneKey& o) const {
        return conv == o.conv && M == o.M && N == o.N && K == o.K && act == o.act && hasr == o.hasr && Hin == o.Hin && Win == o.Win && Cin == o.Cin &&
               stride == o.stride && Hup == o.Hup;
    }
};
struct TuneKeyHash {
    size_t operator()(const TuneKey& k) const {
        size_t h = 1469598103934665603ull;
        const int v[11] = {k.conv, k.M, k.N, k.K, k.act, k.hasr, k.Hin, k.Win, k.Cin, k.stride, k.Hup};
        for (int x : v) { h ^= (size_t)(unsigned)x; h *= 1099511628211ull; }
        return h;
    }
};
static std::unordered_map<TuneKey, int, TuneKeyHash> g_tune_cache;
static std::unordered_map<TuneKey, int, TuneKeyHash> g_near_cache;      // un-tabled shapes -> the tile of their nearest-M twin (0 = none)
#define TUNE_MAGIC "!tcl-gemm-table gfx950 v4"        // bump when tiles / schedules change: older tables are refused, not trusted
static int g_autotune = 1;

static int split_rule(int M, int N, int K, int act) {
    const int t128 = cdiv(M, 128) * cdiv(N, 128), nk = K / 32;
    if (act == 2 || !g_ws || t128 >= 384 || nk < 64) return 1;
    int s = (640 + t128 / 2) / t128;
    if (s > 8) s = 8;
    if (s > nk / 16) s = nk / 16;
    while (s > 1 && (size_t)s * M * N * 4 > g_ws_bytes) --s;
    return s < 1 ? 1 : s;
}

static int heuristic_cfg(int M, int N, int K, int ldc, int ldr, bool has_resid, int act, const ConvP& cp) {
    if (cfg_ok(1, M, N, K, ldc, ldr, has_resid, act, cp)) return (N % 128 == 0 || N > 192) ? 1 : 3;
    return (N % 128 == 0 || N > 512) ? 9 : 10;
}

static int g_near_on = -1;      // un-tabled shapes take their nearest-M twin's tile (TCL_GEMM_NEAR, re-read by tcl_gemm_autotune / tcl_gemm_tune_load)
static void read_near_env() { const char* e = getenv("TCL_GEMM_NEAR"); g_near_on = !(e && atoi(e) == 0); }

static int dispatch(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                    int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st, TclProfScope* ps = nullptr) {
    if (act == 2 && (N % 64 != 0 || (ldc & 7) || resid)) return TCL_EINVAL;      // GEGLU epilogue: 64-column [value | gate] groups
    if (g_tune_cfg) {
        if (!cfg_ok(g_tune_cfg, M, N, K, ldc, ldr, resid != nullptr, act, cp)) return TCL_EINVAL;
        return run_cfg(g_tune_cfg, g_tune_splits > 0 ? g_tune_splits : 1, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    }
    const bool hasr = resid != nullptr;
    const int splits = split_rule(M, N, K, act);
    const int fallback = heuristic_cfg(M, N, K, ldc, ldr, hasr, act, cp);
    if (fallback >= 9) return run_cfg(fallback, 1, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);   // odd N / ld: register-staged kernel
    // re-running the call must be idempotent: no tuning when the output overlaps an input
    const char* c0 = (const char*)C; const char* c1 = c0 + ((size_t)(M - 1) * ldc + N) * 2;
    auto overlaps = [&](const void* p, size_t bytes) { return p && (const char*)p < c1 && (const char*)p + bytes > c0; };
    const size_t a_bytes = cp.conv ? (size_t)(M / (cp.Hout * cp.Wout)) * cp.Hin * cp.Win * cp.Cin * 2 : ((size_t)(M - 1) * lda + K) * 2;
    if (!g_autotune || overlaps(A, a_bytes) || (resid && overlaps(resid, ((size_t)(M - 1) * ldr + N) * 2)))
        return run_cfg(fallback, splits, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    const TuneKey key = {cp.conv, M, N, K, act, (int)hasr, cp.Hin, cp.Win, cp.Cin, cp.stride, cp.Hup};
    auto it = g_tune_cache.find(key);
    if (it != g_tune_cache.end()) {
        // the key leaves the leading dimensions out: a cached (or file-loaded) tile is re-validated for this call's ldc / ldr
        const int c = (cfg_ok(it->second, M, N, K, ldc, ldr, hasr, act, cp) && tile_ok(it->second, M, N, K, splits)) ? it->second : fallback;
        return run_cfg(c, splits, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    }
    // A shape the table does not hold, but whose twin with another row count it does (the same layer in a pass over another number of frames: a
    // group cut at a chunk boundary, the last window of a clip), takes the twin's tile when the row counts are within 2.5x of each other: the best
    // tile moves little with M at these sizes, timing costs a host sync and ~40 launches, and every tile yields the same bits.  The guess lives in
    // its own cache (tcl_gemm_tune_save writes measured entries only: with the timed tuner on, a shape that has such a twin is deliberately NOT measured
    // -- a pass over a clip whose chunk groups change length every step would otherwise time ~40 shapes per step inside the pass).  TCL_GEMM_NEAR=0: off
    // (the re-tuning tools measure every shape); the variable is read again whenever tcl_gemm_autotune / tcl_gemm_tune_load is called.
    if (g_near_on < 0) read_near_env();
    if (g_near_on) {
        auto nt = g_near_cache.find(key);
        int c = nt != g_near_cache.end() ? nt->second : -1;
        if (c < 0) {
            double bd = 1e30;
            c = 0;
            for (const auto& kv : g_tune_cache) {
                TuneKey k = kv.first;
                const int m = k.M;
                k.M = M;
                if (!(k == key)) continue;
                const double dd = fabs(log((double)m / (double)M));
                if (dd < bd) { bd = dd; c = kv.second; }
            }
            if (bd > log(2.5)) c = 0;
            g_near_cache[key] = c;
        }
        if (c > 0 && cfg_ok(c, M, N, K, ldc, ldr, hasr, act, cp) && tile_ok(c, M, N, K, splits))
            return run_cfg(c, splits, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    }
    if (g_autotune == 2)      // table-only mode: shapes the loaded table does not know take the static heuristic (no timing, no host sync)
        return run_cfg(fallback, splits, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    // candidates: LDS-DMA tiles always; the 8-wave kernels when there is no K split and enough tiles to occupy the CUs (tile_ok)
    if (ps) ps->cancel();         // a tuned call is not a launch: keep its ~40 timed runs and host syncs out of the roofline brackets
    int cand[13], nc = 0;
    static const int all_cfgs[13] = {1, 2, 3, 4, 11, 5, 6, 7, 8, 12, 13, 14, 15};
    for (int c : all_cfgs)
        if (tile_ok(c, M, N, K, splits)) cand[nc++] = c;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // two interleaved rounds, minimum per candidate: one disturbed measurement (clock ramp, a profiler attached) must not pick the tile
    float t_ms[13]; bool ok[13];
    for (int i = 0; i < nc; ++i) {
        t_ms[i] = 1e30f;
        ok[i] = cfg_ok(cand[i], M, N, K, ldc, ldr, hasr, act, cp) &&
                run_cfg(cand[i], splits, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st) == TCL_OK;                 // warm-up
    }
    for (int round = 0; round < 2; ++round)
        for (int i = 0; i < nc; ++i) {
            if (!ok[i]) continue;
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < 3; ++r) run_cfg(cand[i], splits, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            t_ms[i] = fminf(t_ms[i], ms);
        }
    int best = fallback; float best_ms = 1e30f;
    for (int i = 0; i < nc; ++i) if (ok[i] && t_ms[i] < best_ms) { best_ms = t_ms[i]; best = cand[i]; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    g_tune_cache[key] = best;
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;      // the timed runs already produced C
}

extern "C" {

int tcl_set_workspace(void* ws, size_t bytes) { g_ws = (float*)ws; g_ws_bytes = ws ? bytes : 0; return TCL_OK; }
int tcl_gemm_tune(int cfg, int splits) { g_tune_cfg = cfg; g_tune_splits = splits; return TCL_OK; }
int tcl_gemm_autotune(int enable) { g_autotune = enable; g_near_cache.clear(); read_near_env(); if (!enable) g_tune_cache.clear(); return TCL_OK; }

// Persistent tuning table: one text line per problem shape, "conv M N K act hasr Hin Win Cin stride Hup cfg".
int tcl_gemm_tune_save(const char* path) {
    TCL_CHECK_ARG(path);
    FILE* f = fopen(path, "w");
    if (!f) return TCL_EINVAL;
    fprintf(f, "# tc_light_amd GEMM tile table, gfx950: conv M N K act hasr Hin Win Cin stride Hup -> cfg (csrc/gemm.hip)\n");
    fprintf(f, "%s\n", TUNE_MAGIC);
    for (const auto& kv : g_tune_cache) {
        const TuneKey& k = kv.first;
        fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d %d\n", k.conv, k.M, k.N, k.K, k.act, k.hasr, k.Hin, k.Win, k.Cin, k.stride, k.Hup, kv.second);
    }
    fclose(f);
    return TCL_OK;
}
int tcl_gemm_tune_load(const char* path) {
    TCL_CHECK_ARG(path);
    FILE* f = fopen(path, "r");
    if (!f) return TCL_EINVAL;
    char line[256];
    bool versioned = false;
    while (fgets(line, sizeof line, f)) {
        TuneKey k; int cfg;
        if (!strncmp(line, TUNE_MAGIC, strlen(TUNE_MAGIC))) { versioned = true; continue; }
        if (line[0] == '#') continue;
        if (!versioned) { fclose(f); return TCL_EINVAL; }          // a table of another kernel generation / architecture: measured with other tiles
        if (sscanf(line, "%d %d %d %d %d %d %d %d %d %d %d %d", &k.conv, &k.M, &k.N, &k.K, &k.act, &k.hasr, &k.Hin, &k.Win, &k.Cin, &k.stride, &k.Hup, &cfg) != 12) continue;
        if (!((cfg >= 1 && cfg <= 8) || (cfg >= 11 && cfg <= 15))) continue;         // only ids the cached path may run
        g_tune_cache[k] = cfg;
    }
    fclose(f);
    g_near_cache.clear();
    read_near_env();
    return TCL_OK;
}
size_t tcl_gemm_tune_size(void) { return g_tune_cache.size(); }

int tcl_gemm_f16(const void* A, const void* W, const void* bias, const void* resid, void* C, int M, int N, int K, int lda, int ldw,
                 int ldc, int ldr, int act, hipStream_t st) {
    TCL_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0 && K % BK == 0 && lda % 8 == 0 && lda >= K && ldw % 8 == 0 && ldw >= K && act >= 0 && act <= 5);
    ConvP cp = {};
    TclProfScope ps(TCL_PROF_GEMM, st, 2.0 * M * N * K);
    return dispatch((const _Float16*)A, (const _Float16*)W, (const _Float16*)bias, (const _Float16*)resid, (_Float16*)C, M, N, K, lda,
                    ldw, ldc, ldr, act, cp, st, &ps);
}

int tcl_conv3x3_f16(const void* X, const void* W, const void* bias, const void* resid, void* Y, int B, int Hin, int Win, int Cin,
                    int Cout, int stride, int pad, int Hup, int Wup, int act, hipStream_t st) {
    TCL_CHECK_ARG(X && W && Y && B > 0 && Cin % BK == 0 && Cout > 0 && (stride == 1 || stride == 2) && (pad == 0 || pad == 1));
    ConvP cp;
    cp.conv = 1; cp.Hin = Hin; cp.Win = Win; cp.Cin = Cin;
    cp.Hup = Hup > 0 ? Hup : Hin; cp.Wup = Wup > 0 ? Wup : Win;
    cp.stride = stride; cp.pad = pad;
    // pad=1: k3 s1|s2 p1 (UNet);  pad=0 & stride 2: the VAE encoder's asymmetric (0,1,0,1) padding
    cp.Hout = pad ? (cp.Hup + 2 - 3) / stride + 1 : (cp.Hup + 1 - 3) / stride + 1;
    cp.Wout = pad ? (cp.Wup + 2 - 3) / stride + 1 : (cp.Wup + 1 - 3) / stride + 1;
    cp.sy = (float)Hin / (float)cp.Hup; cp.sx = (float)Win / (float)cp.Wup;
    const int M = B * cp.Hout * cp.Wout;
    TclProfScope ps(TCL_PROF_GEMM, st, 2.0 * M * Cout * 9.0 * Cin);
    return dispatch((const _Float16*)X, (const _Float16*)W, (const _Float16*)bias, (const _Float16*)resid, (_Float16*)Y, M, Cout,
                    9 * Cin, 0, 9 * Cin, Cout, Cout, act, cp, st, &ps);
}

}  // extern "C"
