// Short-K Linear layers (K = 320: q / k / v / out projections, proj_in / proj_out, the GEGLU feed-forward of the level-0 transformer
// blocks: generate.py:342-347 -> diffusers BasicTransformerBlock), strip-resident form.
// A tiled GEMM spends a K = 320 problem in its prologue and epilogue (ten 32-wide K steps per tile: matrix pipe 25 % busy, 1.5-2.4 TB/s,
// profiles/r3_gemm_dma_counters.txt).  Here the decomposition is the one of the VidToMe matching kernel (merge.hip::k_tome_match320,
// matrix pipe 54 %): a block OWNS a strip of 128 activation rows -- each wave keeps the MFMA B operand of its 32 rows in registers for
// ALL of K (20 half8 = 80 VGPRs), read from HBM exactly once -- and SWEEPS the weight rows in tiles of 128, which stream from L2
// through a 4-slot LDS ring by LDS-DMA (64-wide K stages, 128 B per row, 16-B chunks XOR-swizzled).  The sweep is one long pipeline
// (N = 2560: 100 stages), the epilogue of a tile overlaps the next tile's DMA, and C leaves in 16-B pieces straight from registers:
// the two lanes that share an output row exchange half of their packed results with v_permlane32_swap, no LDS staging.
// Results are bit-identical to every other tile configuration of csrc/gemm.hip: same MFMA (32x32x16 f16), same K order, same
// epilogue arithmetic (f16(act(acc + bias)), then f16(post_act(. + resid)); GEGLU on the [32 value | 32 gate] row groups of unet.py::_geglu_rows).
#include "common.h"
#include "gemm_conv.h"
#include "prof.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// 16 results of a lane (one 32 x 32 MFMA tile: row n = 8 (r >> 2) + 4 hl + (r & 3), column = the lane's activation row) -> the two 16-B
// pieces the lane stores: columns [8 hl, 8 hl + 8) and [16 + 8 hl, 16 + 8 hl + 8) of the tile's 32 output columns.
__device__ __forceinline__ void ls_exchange(const _Float16 (&o)[16], uint4v& p0, uint4v& p1) {
    unsigned d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { half2v h = {o[2 * q], o[2 * q + 1]}; d[q] = __builtin_bit_cast(unsigned, h); }
    // lane L < 32 keeps (d0, d1) and takes the partner's (d0, d1); lane L + 32 takes L's (d2, d3) and keeps its own
    const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
    const auto s2 = __builtin_amdgcn_permlane32_swap(d[4], d[6], false, false);
    const auto s3 = __builtin_amdgcn_permlane32_swap(d[5], d[7], false, false);
    p0 = uint4v{s0[0], s1[0], s0[1], s1[1]};
    p1 = uint4v{s2[0], s3[0], s2[1], s3[1]};
}

// Q-panel epilogue (round 5): instead of C [M, N] the result goes straight into the flash kernel's query panel Qp [B, H, Tqp, 48] (csrc/attn.hip:
// head-major rows of DP = 48 halves, pre-scaled by softmax_scale * log2 e, columns 40..47 zero) -- the attn2 to_q projection of the C = 320
// transformer blocks then needs neither its [M, 320] output nor the pack pass that re-read it (k_pack_rows: 1.07 s per 300-frame pass).  Rows
// t >= Tq of the panel are never written here: the caller keeps them zero.  Same rounding points as Linear -> pack: f16(acc), then f16(. * scale).
struct QPanel { _Float16* panel; int Tq, Tqp; float scale; };
#ifndef LS_ABL
#define LS_ABL 0        // lab-only knock-outs (tools/micro/lin_lab.hip): 1 no C stores, 2 no A loads, 4 no residual loads, 8 no sweep (DMA + MFMA)
#endif
// LN: the rows of A are LayerNorm-ed (gamma, beta, eps over the K = C elements) on their way into the B-operand registers -- the row is already
// spread over the two lanes that hold it, so the statistics cost one lane exchange and the normalised activations never exist in memory
// (tcl_ln_gemm_f16: norm2 -> to_q of attn2, norm3 -> GEGLU feed-forward of the C = 320 transformer blocks).
template <int K, int NW, bool LN>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_lin_strip(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                                       const _Float16* __restrict__ bias, const _Float16* __restrict__ resid,
                                                                       _Float16* __restrict__ C, int M, int N, int lda, int ldw, int ldc, int ldr,
                                                                       int act, int tiles_n, int nsplit, const _Float16* __restrict__ gamma,
                                                                       const _Float16* __restrict__ beta, float eps, QPanel qp) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NST = K / 64, STAGE = 128 * 128, SW = 32 * NW, NP = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* const lds0 = (__attribute__((address_space(3))) char*)smem;
    _Float16* const sbias = (_Float16*)(smem + 4 * STAGE);                       // the block's bias range, 128 entries per tile
    const int bid = blockIdx.x, split = bid % nsplit, strip = bid / nsplit;     // the splits of a strip are neighbours: its rows are fetched once into L2
    const int tps = (tiles_n + nsplit - 1) / nsplit, t0 = split * tps, t1 = min(t0 + tps, tiles_n);
    if (t0 >= t1) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), hl = lane >> 5, col = lane & 31;
    const int ntl = t1 - t0, nstep = ntl * NST;
    // DMA roles (as k_tome_match320): wave w stages pieces NP w .. NP w + NP - 1 of a stage (8 weight rows x 128 B each); a piece's source is a
    // scalar base plus one of two per-lane byte offsets (the swizzle term of row R = 8 piece + rr only depends on the piece's parity)
    const int rr = lane >> 3, ch = lane & 7;
    const int voff0 = rr * (ldw * 2) + ((ch ^ (rr >> 1)) << 4), voff1 = rr * (ldw * 2) + ((ch ^ (4 + (rr >> 1))) << 4);
    const int m = strip * SW + wid * 32 + col;                                   // this lane's activation row = output row
    const bool live = m < M;
    const int q_b = qp.panel ? m / qp.Tq : 0, q_t = qp.panel ? m - q_b * qp.Tq : 0, q_H = N / 40;      // panel mode: (sample, token) of the row
    half8 bfr[K / 16];                                                            // B operand: row m, k = 16 ks + 8 hl .. + 7
    {
        const _Float16* ap = A + (long)(live ? m : 0) * lda + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            if (LS_ABL & 2) { for (int j = 0; j < 8; ++j) bfr[ks][j] = (_Float16)(float)(lane + ks + j); }
            else bfr[ks] = *(const half8*)(ap + ks * 16);
        }
    }
    int i_t = 0, i_k = 0;                                                         // (tile, stage) of the step being ISSUED
#define LS_ISSUE(BUF)                                                                                                         \
    {                                                                                                                         \
        const int n0_ = min((t0 + i_t) * 128, N - 128);                                                                       \
        const char* sb_ = (const char*)W + ((long)(n0_ + wid * (8 * NP)) * ldw + i_k * 64) * 2;                               \
        _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                                     \
            const char* src_ = sb_ + (long)i * (8 * ldw * 2) + ((i & 1) ? voff1 : voff0);                                     \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                             \
                                             (__attribute__((address_space(3))) void*)(lds0 + (BUF) * STAGE + (wid * NP + i) * 1024), 16, 0, 0); \
        }                                                                                                                     \
        if (++i_k == NST) { i_k = 0; ++i_t; }                                                                                 \
    }
    // the first two weight stages are requested before anything waits for the activation rows: their L2 latency overlaps the strip's HBM latency
    if (!(LS_ABL & 8)) { LS_ISSUE(0); if (nstep > 1) LS_ISSUE(1); }
    for (int i = tid; i < ntl * 128; i += 64 * NW) {                              // bias of the swept tiles (the last tile is moved back, like its rows)
        const int n = min((t0 + i / 128) * 128, N - 128) + (i & 127);
        sbias[i] = bias ? bias[n] : (_Float16)0.f;
    }
    if (LN) {       // torch.nn.LayerNorm over the row: f32 statistics (two passes), f16 result -- under the latency of the first weight stages
        float s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) s1 += (float)bfr[ks][j];
        const float o1 = __shfl_xor(s1, 32, 64);
        const float mean = (hl ? o1 + s1 : s1 + o1) / K;                          // lower half + upper half on both lanes
        float s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = (float)bfr[ks][j] - mean; s2 += d * d; }
        const float o2 = __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf((hl ? o2 + s2 : s2 + o2) / K + eps);
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            const half8 g = *(const half8*)(gamma + ks * 16 + 8 * hl), bt = *(const half8*)(beta + ks * 16 + 8 * hl);
#pragma unroll
            for (int j = 0; j < 8; ++j) bfr[ks][j] = (_Float16)(((float)bfr[ks][j] - mean) * rstd * (float)g[j] + (float)bt[j]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // bias in LDS before the first barrier
    int step = 0;
    // (Requesting a tile's residual pieces at the START of its K loop was measured: 269 us against 248 at 368 640 x 320 -- the vmcnt(0) of
    // the next ring barrier waits for them at once.  They are read in the epilogue.)
    const bool has_r = resid != nullptr && act != 2 && !(LS_ABL & 4);
    for (int tl = 0; tl < ntl; ++tl) {
        float16v acc[4];
        if (LS_ABL & 8) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = (float)bfr[(a * 16 + r) % (K / 16)][r & 7];
        } else
#pragma unroll
        for (int kt = 0; kt < NST; ++kt, ++step) {
            // 4-slot ring, two K stages per barrier: steps s, s+1 (s even) are consumed while s+2, s+3 stream into the slots of s-2, s-1.
            // The epilogue's stores and residual loads are VMEM too: vmcnt(0) at the barrier drains them with the DMA (in order anyway).
            if ((step & 1) == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (step + 2 < nstep) LS_ISSUE((step + 2) & 3);
                if (step + 3 < nstep) LS_ISSUE((step + 3) & 3);
            }
            const char* db = smem + (step & 3) * STAGE;
            half8 fa[2][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { const int R = a * 32 + col; fa[0][a] = *(const half8*)(db + R * 128 + (((0 + hl) ^ ((R >> 1) & 7)) << 4)); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) { const int R = a * 32 + col; fa[(ks + 1) & 1][a] = *(const half8*)(db + R * 128 + (((2 * (ks + 1) + hl) ^ ((R >> 1) & 7)) << 4)); }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (kt == 0 && ks == 0) {
                        float16v z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][a], bfr[0], z, 0, 0, 0);
                    } else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][a], bfr[kt * 4 + ks], acc[a], 0, 0, 0);
                }
            }
        }
        // ---- epilogue of the tile: 128 output columns [n0, n0 + 128) of the lane's row (64 with GEGLU)
        const int n0 = min((t0 + tl) * 128, N - 128);
        const _Float16* bt = sbias + tl * 128 + 4 * hl;
        if (act == 2) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {                                      // rows [64 g2, +32) value, [64 g2 + 32, +32) the matching gates
                _Float16 o[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const half4 bv = *(const half4*)(bt + g2 * 64 + 8 * q), bg = *(const half4*)(bt + g2 * 64 + 32 + 8 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const _Float16 va = (_Float16)(acc[2 * g2][4 * q + j] + (float)bv[j]), vg = (_Float16)(acc[2 * g2 + 1][4 * q + j] + (float)bg[j]);
                        o[4 * q + j] = (_Float16)((float)va * gelu_erf((float)vg));
                    }
                }
                uint4v p0, p1;
                ls_exchange(o, p0, p1);
                if (live && (!(LS_ABL & 1) || p0[0] == 0x12345678u)) {
                    _Float16* cp = C + (long)m * ldc + (n0 >> 1) + g2 * 32 + 8 * hl;
                    *(uint4v*)cp = p0;
                    *(uint4v*)(cp + 16) = p1;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                _Float16 o[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const half4 bv = *(const half4*)(bt + a * 32 + 8 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * q + j] = (_Float16)apply_act(acc[a][4 * q + j] + (float)bv[j], act);
                }
                uint4v p0, p1;
                ls_exchange(o, p0, p1);
                if (live && qp.panel) {
                    half8 v[2] = {__builtin_bit_cast(half8, p0), __builtin_bit_cast(half8, p1)};
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[pc][j] = (_Float16)((float)v[pc][j] * qp.scale);
                        const int c = n0 + a * 32 + 8 * hl + 16 * pc, head = c / 40, dd = c - head * 40;
                        _Float16* dst = qp.panel + (((long)q_b * q_H + head) * qp.Tqp + q_t) * 48 + dd;
                        *(half8*)dst = v[pc];
                        if (dd == 32) { half8 z; for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f; *(half8*)(dst + 8) = z; }      // the panel's padding columns 40..47
                    }
                    continue;
                }
                if (live) {
                    const long off = n0 + a * 32 + 8 * hl;
                    if (has_r) {
                        const half8 r0 = *(const half8*)(resid + (long)m * ldr + off), r1 = *(const half8*)(resid + (long)m * ldr + off + 16);
                        half8 v0 = __builtin_bit_cast(half8, p0), v1 = __builtin_bit_cast(half8, p1);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            v0[j] = (_Float16)post_act((float)v0[j] + (float)r0[j], act);
                            v1[j] = (_Float16)post_act((float)v1[j] + (float)r1[j], act);
                        }
                        p0 = __builtin_bit_cast(uint4v, v0); p1 = __builtin_bit_cast(uint4v, v1);
                    }
                    _Float16* cp = C + (long)m * ldc + off;
                    if ((LS_ABL & 1) && p0[0] != 0x12345678u) continue;
                    *(uint4v*)cp = p0;
                    *(uint4v*)(cp + 16) = p1;
                }
            }
        }
    }
#undef LS_ISSUE
#endif
}


int lin_strip_dispatch_q(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                         int lda, int ldw, int ldc, int ldr, int act, hipStream_t st, const _Float16* gamma, const _Float16* beta, float eps, QPanel qp);
// cfg 12 of csrc/gemm.hip.  Valid for dense A, K == 320, N >= 128 and N % 32 == 0 (GEGLU: N % 64 == 0), 16-B aligned rows.
bool lin_strip_ok(int M, int N, int K, int lda, int ldw, int ldc, int ldr, bool has_resid, int act, const ConvP& cp) {
    return !cp.conv && K == 320 && N >= 128 && N % (act == 2 ? 64 : 32) == 0 && N <= 8192 && (lda & 7) == 0 && (ldw & 7) == 0 && (ldc & 7) == 0 &&
           (!has_resid || (ldr & 7) == 0) && (long)ldw * 2 * 8 < (1l << 30);
}

int lin_strip_dispatch(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                       int lda, int ldw, int ldc, int ldr, int act, hipStream_t st, const _Float16* gamma, const _Float16* beta, float eps) {
    return lin_strip_dispatch_q(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, st, gamma, beta, eps, QPanel{nullptr, 1, 1, 1.f});
}
int lin_strip_dispatch_q(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                         int lda, int ldw, int ldc, int ldr, int act, hipStream_t st, const _Float16* gamma, const _Float16* beta, float eps, QPanel qp) {
    if (K != 320) return TCL_EINVAL;
    constexpr int NW = 4;
    const int tn = cdiv(N, 128), strips = cdiv(M, 32 * NW);
    // enough blocks for ~3 rounds of the 512 slots (2 per CU); a split re-reads the strip (from L2) and sweeps its share of the weight tiles
    static const int force_split = getenv("TCL_LS_SPLIT") ? atoi(getenv("TCL_LS_SPLIT")) : 0;      // lab hook
    int nsplit = 1;
    while (nsplit < 8 && (long)strips * nsplit < 512 * 2 && tn / (nsplit * 2) >= 2) nsplit *= 2;
    if (force_split > 0 && tn / force_split >= 1) nsplit = force_split;
    const size_t lds = (size_t)4 * 128 * 128 + (size_t)cdiv(tn, nsplit) * 128 * 2;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute((const void*)k_lin_strip<320, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 128 + 64 * 128 * 2);
        (void)hipFuncSetAttribute((const void*)k_lin_strip<320, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 128 + 64 * 128 * 2);
        set = true;
    }
    if (gamma) hipLaunchKernelGGL((k_lin_strip<320, NW, true>), dim3(strips * nsplit), dim3(64 * NW), lds, st, A, W, bias, resid, C, M, N, lda, ldw, ldc, ldr, act, tn, nsplit, gamma, beta, eps, qp);
    else hipLaunchKernelGGL((k_lin_strip<320, NW, false>), dim3(strips * nsplit), dim3(64 * NW), lds, st, A, W, bias, resid, C, M, N, lda, ldw, ldc, ldr, act, tn, nsplit, gamma, beta, eps, qp);
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

extern "C" {
#include "../../include/tclight_hip.h"
int tcl_ln_gemm_f16(const void* x, const void* gamma, const void* beta, float eps, const void* W, const void* bias, const void* resid, void* C,
                    int M, int N, int K, int ldx, int ldw, int ldc, int ldr, int act, hipStream_t st) {
    TCL_CHECK_ARG(x && gamma && beta && W && C && M > 0 && act >= 0 && act <= 5 && (act != 2 || !resid));
    ConvP cp = {};
    TCL_CHECK_ARG(lin_strip_ok(M, N, K, ldx, ldw, ldc, ldr, resid != nullptr, act, cp) && ldx >= K && ldw >= K);
    TCL_CHECK_ARG(!(resid == C && N % 128 != 0));      // no in-place residual when the last weight tile is moved back (its columns are visited twice)
    TclProfScope ps(TCL_PROF_GEMM, st, 2.0 * M * N * K);
    return lin_strip_dispatch((const _Float16*)x, (const _Float16*)W, (const _Float16*)bias, (const _Float16*)resid, (_Float16*)C, M, N, K, ldx, ldw, ldc,
                              ldr, act, st, (const _Float16*)gamma, (const _Float16*)beta, eps);
}
int tcl_ln_gemm_qpanel_f16(const void* x, const void* gamma, const void* beta, float eps, const void* W, int M, int H, int d, int Tq, int ldx, int ldw,
                           float scale, void* ws_q, hipStream_t st) {
    TCL_CHECK_ARG(x && gamma && beta && W && ws_q && M > 0 && H > 0 && d == 40 && Tq > 0 && M % Tq == 0);
    const int N = H * d, K = N;
    ConvP cp = {};
    TCL_CHECK_ARG(lin_strip_ok(M, N, K, ldx, ldw, N, N, false, 0, cp) && ldx >= K && ldw >= K);
    TclProfScope ps(TCL_PROF_GEMM, st, 2.0 * M * N * K);
    const QPanel qp = {(_Float16*)ws_q, Tq, (Tq + 255) / 256 * 256, scale * 1.4426950408889634f};
    return lin_strip_dispatch_q((const _Float16*)x, (const _Float16*)W, nullptr, nullptr, (_Float16*)ws_q, M, N, K, ldx, ldw, N, N, 0, st,
                                (const _Float16*)gamma, (const _Float16*)beta, eps, qp);
}
}
