// Flash attention forward for gfx950 (f16 in, f32 softmax/accumulate), head_dim 40 / 80 / 160 (SD-1.5 UNet) --
// the self-attention over VidToMe-merged tokens (T up to ~47k) and the text cross-attention that the reference runs
// through torch SDPA / xformers (AttnProcessor2_0, utils/model_utils.py:66-67; patch.py:170-176).  T x T scores are
// never materialised.
//
// Two kernels:
//  pack : Q,K,V rows (heads interleaved in channels) -> head-major, zero-padded panels
//           Qp [B,H,Tqp,DP] (pre-scaled by softmax_scale*log2 e), Kp [B,H,Tkp,DP], Vt [B,H,DPV,Tkp] (V transposed so the
//           PV contraction reads keys contiguously).  DP = d rounded to 16, DPV = d rounded to 32.
//  flash: block = 4 waves x 32 query rows; 64-key K/V tiles staged global->VGPR->LDS, double-buffered.
//         S^T = K.Q^T with mfma_f32_32x32x16_f16 (swapped operands: lane l owns query l&31, so the row max/sum is
//         in-lane + one lane<->lane+32 exchange); P stays in registers as the B operand of O^T = V^T.P^T (the key
//         permutation of the accumulator layout is matched by the V^T fragment addresses instead of shuffling P).
//         Block id -> head = id % H so that each head's K/V panel lives in one XCD's L2.
#include "common.h"
#include "../../include/tclight_hip.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define KV_TILE 64
#define V_STRIDE 72   // halves: 144 B = 9 x 16 B (odd) -> the 16-lane groups of a ds_read_b128 are conflict-free

__global__ void k_pack_rows(const _Float16* __restrict__ src, long bstride, int ld, int T, int H, int d, float scale,
                            _Float16* __restrict__ dst, int Tp, int DP, long total_chunks) {
    const int cpr = DP / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
        int c8 = (int)(i % cpr) * 8; long row = i / cpr; int t = (int)(row % Tp); long bh = row / Tp; int h = (int)(bh % H); long b = bh / H;
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
        if (t < T && c8 < d) {
            v = *(const half8*)(src + b * bstride + (long)t * ld + h * d + c8);
            if (scale != 1.f)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (_Float16)((float)v[j] * scale);
        }
        *(half8*)(dst + i * 8) = v;
    }
}
// Vt[b][h][i][p(t)] = V[b][t][h*d + i]; one block per (64 tokens, b*h).  Within every group of 16 keys the two middle blocks of 4
// are swapped (p swaps bits 2 and 3 of t): the S^T accumulator of the flash kernel leaves lane half hl with keys {4hl..4hl+3,
// 8+4hl..8+4hl+3} of a group, and with this order those 8 V^T values are one contiguous 16-B LDS read (8*hl .. 8*hl+7).
__global__ __launch_bounds__(256) void k_pack_vt(const _Float16* __restrict__ v, long bstride, int ld, int T, int H, int d,
                                                 _Float16* __restrict__ vt, int Tp, int DPV) {
    extern __shared__ _Float16 tile[];   // [64][DPV+2]
    const int t0 = blockIdx.x * 64, bh = blockIdx.y, h = bh % H; const long b = bh / H;
    const int st = DPV + 2, cpr = DPV / 8;
    for (int i = threadIdx.x; i < 64 * cpr; i += 256) {
        int r = i / cpr, c8 = (i % cpr) * 8, t = t0 + r;
        half8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (_Float16)0.f;
        if (t < T && c8 < d) x = *(const half8*)(v + b * bstride + (long)t * ld + h * d + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[r * st + c8 + j] = x[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DPV * 64; i += 256) {
        int dd = i / 64, r = i % 64;
        _Float16 val = tile[r * st + dd];
        if (dd == d && DPV > d) val = (t0 + r < T) ? (_Float16)1.f : (_Float16)0.f;   // ones row: the PV MFMA also yields the softmax row sums
        const int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
        vt[((long)bh * DPV + dd) * Tp + t0 + pr] = val;
    }
}

// One 64-key tile for one wave (32 queries): S^T = K.Q^T, online softmax, O^T += V^T.P^T.
// LROW: the row sums l come out of the PV MFMA itself through a ones-row that k_pack_vt stores at Vt row D (free padding
// row when DPV > D), so no VALU adds are spent on them.  Rescaling of O is lazy: only when some query of the wave sees its
// running max grow by more than 2^6 (wave-uniform branch); P then stays <= 64, far inside f16 range.
template <int DP, int DPV, bool LROW>
__device__ __forceinline__ void flash_tile(const _Float16* __restrict__ kt, const _Float16* __restrict__ vt, const half8 (&qf)[DP / 16],
                                           float16v (&o)[DPV / 32], float& m, float& lsum, int ql, int hl, int kv0, int Tk, bool mask) {
    constexpr int KS = DP + 8, NQK = DP / 16, NDT = DPV / 32;
    float16v s[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
        const _Float16* kr = kt + (blk * 32 + ql) * KS + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NQK; ++ks) s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)(kr + ks * 16), qf[ks], s[blk], 0, 0, 0);
    }
    if (__builtin_amdgcn_readfirstlane((int)mask)) {      // scalar branch: only the last tile has padded keys
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) { int kv = kv0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl; if (kv >= Tk) s[blk][r] = -1e30f; }
    }
    float mx = s[0][0];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (__any(mx > m + 6.f)) {                     // lazy rescale (rare after the first tiles)
        const float mn = fmaxf(m, mx), alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        lsum *= alpha;
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    }
    half8 pf[2][2];
    float ps = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) { float p = __builtin_amdgcn_exp2f(s[blk][r] - m); if (!LROW) ps += p; pf[blk][r >> 3][r & 7] = (_Float16)p; }
    if (!LROW) lsum += ps;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const _Float16* vr = vt + ql * V_STRIDE + blk * 32 + 16 * ss + 8 * hl;
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
                half8 vf = *(const half8*)(vr + t * 32 * V_STRIDE);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[blk][ss], o[t], 0, 0, 0);
            }
        }
}

template <int D, int DP, int DPV, int NW>
__global__ __launch_bounds__(NW * 64, (DP <= 48 && NW == 4) ? 3 : 2) void k_flash(const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vt,
                                                  _Float16* __restrict__ O, int H, int Tq, int Tk, int Tqp, int Tkp, int d, int ldo, long obstride,
                                                  int kv_div, int nqb) {
    constexpr int KS = DP + 8;                    // K row stride (halves); (DP+8)/8 odd -> conflict-free b128 reads
    constexpr int NQK = DP / 16, NDT = DPV / 32;
    constexpr int KCH = KV_TILE * DP / 8, VCH = DPV * 8;           // 16-B chunks per tile
    constexpr int NT_ = NW * 64;                                   // threads per block (NW waves x 32 query rows each)
    constexpr int KIT = (KCH + NT_ - 1) / NT_, VIT = (VCH + NT_ - 1) / NT_;
    constexpr bool LROW = DPV > D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* Ks = (_Float16*)smem;                         // [2][64][KS]
    _Float16* Vs = Ks + 2 * KV_TILE * KS;                   // [2][DPV][V_STRIDE]

    const int bid = blockIdx.x, head = bid % H, qb = (bid / H) % nqb, b = bid / (H * nqb);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hl = lane >> 5, ql = lane & 31;
    const int q0 = qb * (NW * 32) + wid * 32;
    const long bh = (long)b * H + head, kbh = (long)(b / kv_div) * H + head;
    const _Float16* kbase = Kp + kbh * Tkp * DP;
    const _Float16* vbase = Vt + kbh * DPV * Tkp;

    half8 qf[NQK];
    {
        const _Float16* qrow = Qp + (bh * Tqp + q0 + ql) * DP + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NQK; ++ks) qf[ks] = *(const half8*)(qrow + ks * 16);
    }
    u32x4 rkA[KIT], rvA[VIT], rkB[KIT], rvB[VIT];      // two staging sets: tile it+1 waits in one while tile it+2 is in flight
#define FLASH_GLOAD(IT, RK, RV)                                                                                               \
    {                                                                                                                         \
        const _Float16* kt_ = kbase + (long)(IT) * KV_TILE * DP;                                                             \
        _Pragma("unroll") for (int i = 0; i < KIT; ++i) { int c = min(tid + NT_ * i, KCH - 1); RK[i] = *(const u32x4*)(kt_ + c * 8); }   /* clamped: always defined */ \
        _Pragma("unroll") for (int i = 0; i < VIT; ++i) { int c = min(tid + NT_ * i, VCH - 1); RV[i] = *(const u32x4*)(vbase + (long)(c >> 3) * Tkp + (IT) * KV_TILE + (c & 7) * 8); } \
    }
#define FLASH_SSTORE(BUF, RK, RV)                                                                                             \
    {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < KIT; ++i) { int c = tid + NT_ * i; if ((i + 1) * NT_ <= KCH || c < KCH) { int r = c / (DP / 8), c8 = (c % (DP / 8)) * 8; *(u32x4*)(Ks + ((BUF) * KV_TILE + r) * KS + c8) = RK[i]; } } \
        _Pragma("unroll") for (int i = 0; i < VIT; ++i) { int c = tid + NT_ * i; if ((i + 1) * NT_ <= VCH || c < VCH) { *(u32x4*)(Vs + ((BUF) * DPV + (c >> 3)) * V_STRIDE + (c & 7) * 8) = RV[i]; } } \
    }
#define FLASH_TILE(BUF, IT) flash_tile<DP, DPV, LROW>(Ks + (BUF) * KV_TILE * KS, Vs + (BUF) * DPV * V_STRIDE, qf, o, m, lsum, ql, hl, (IT) * KV_TILE, Tk, (IT) >= nfull)

    float16v o[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -1e30f, lsum = 0.f;

    const int nt = Tkp / KV_TILE, nfull = Tk / KV_TILE;      // tiles without padded keys
    constexpr bool PF2 = DP <= 80;        // prefetch distance 2 (two register sets) where the register budget allows it
    FLASH_GLOAD(0, rkA, rvA); FLASH_SSTORE(0, rkA, rvA);
    if constexpr (PF2) {
        if (nt > 1) FLASH_GLOAD(1, rkA, rvA);
        __syncthreads();
        for (int it = 0; it < nt; it += 2) {
            // even tile in LDS buffer 0; set A holds tile it+1; tile it+2 goes into set B (two iterations to land)
            if (it + 2 < nt) FLASH_GLOAD(it + 2, rkB, rvB);
            FLASH_TILE(0, it);
            if (it + 1 < nt) FLASH_SSTORE(1, rkA, rvA);
            __syncthreads();
            if (it + 1 >= nt) break;
            if (it + 3 < nt) FLASH_GLOAD(it + 3, rkA, rvA);
            FLASH_TILE(1, it + 1);
            if (it + 2 < nt) FLASH_SSTORE(0, rkB, rvB);
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int it = 0; it < nt; ++it) {
            const int cur = it & 1;
            if (it + 1 < nt) FLASH_GLOAD(it + 1, rkA, rvA);
            FLASH_TILE(cur, it);
            if (it + 1 < nt) FLASH_SSTORE(cur ^ 1, rkA, rvA);
            __syncthreads();
        }
    }
#undef FLASH_TILE
#undef FLASH_GLOAD
#undef FLASH_SSTORE
    // ---- epilogue
    float l;
    if (LROW) {      // l sits in O^T row D: tile D/32, register group (D%32)/8 (D%8 == 0), lanes with hl == (D%8)/4 == 0
        constexpr int TL = D / 32, RG = (D % 32) / 8;
        l = o[TL][4 * RG];
        l = __shfl(l, ql, 64);                    // broadcast from the hl == 0 half
    } else {
        l = lsum + __shfl_xor(lsum, 32, 64);
    }
    const float inv = 1.f / l;
    const int q = q0 + ql;
    if (q < Tq) {
        _Float16* orow = O + (long)b * obstride + (long)q * ldo + head * d;
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int dd = t * 32 + 8 * g + 4 * hl;
                if (dd < d) {
                    half4 w = {(_Float16)(o[t][4 * g] * inv), (_Float16)(o[t][4 * g + 1] * inv), (_Float16)(o[t][4 * g + 2] * inv), (_Float16)(o[t][4 * g + 3] * inv)};
                    *(half4*)(orow + dd) = w;
                }
            }
    }
}

// ---- optional in-library timing of the flash kernel (bench.py roofline leg): HIP events recorded on the launch stream
#include <vector>
struct FlashProf { bool on = false; int dfilter = 0; std::vector<hipEvent_t> ev; double flops = 0.0; long launches = 0; };
static FlashProf g_prof;

template <int D, int DP, int DPV, int NW>
static int launch_flash(const _Float16* Qp, const _Float16* Kp, const _Float16* Vt, _Float16* O, int B, int H, int Tq, int Tk, int Tqp, int Tkp,
                        int d, int ldo, long obs, int kv_div, hipStream_t st) {
    const size_t lds = (size_t)2 * KV_TILE * (DP + 8) * 2 + (size_t)2 * DPV * V_STRIDE * 2;
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)k_flash<D, DP, DPV, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    const int nqb = Tqp / (NW * 32);
    const bool prof = g_prof.on && (g_prof.dfilter == 0 || g_prof.dfilter == d);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, st); }
    hipLaunchKernelGGL((k_flash<D, DP, DPV, NW>), dim3(B * H * nqb), dim3(NW * 64), lds, st, Qp, Kp, Vt, O, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, nqb);
    if (prof) { hipEventRecord(e1, st); g_prof.ev.push_back(e0); g_prof.ev.push_back(e1); g_prof.flops += 4.0 * B * H * (double)Tq * Tk * d; g_prof.launches++; }
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

extern "C" {

// Timing of the flash kernel launches (all head dims, or only head_dim == dfilter) with HIP events on their stream.
int tcl_flash_profile_begin(int dfilter) { g_prof.on = true; g_prof.dfilter = dfilter; g_prof.flops = 0.0; g_prof.launches = 0; g_prof.ev.clear(); return TCL_OK; }
// -> total kernel ms, algorithmic FLOPs (4*B*H*Tq*Tk*d per launch) and launch count since begin; synchronises the events.
int tcl_flash_profile_end(double* total_ms, double* total_flops, long* launches) {
    TCL_CHECK_ARG(total_ms && total_flops && launches);
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g_prof.ev.size(); i += 2) {
        float t = 0.f;
        hipEventSynchronize(g_prof.ev[i + 1]);
        hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
        ms += t;
        hipEventDestroy(g_prof.ev[i]); hipEventDestroy(g_prof.ev[i + 1]);
    }
    *total_ms = ms; *total_flops = g_prof.flops; *launches = g_prof.launches;
    g_prof.on = false; g_prof.ev.clear();
    return TCL_OK;
}

// panel sizes: Tqp = ceil128(Tq), Tkp = ceil64(Tk), DP = ceil16(d), DPV = ceil32(d)
size_t tcl_attention_q_bytes(int B, int H, int Tq, int d) { return (size_t)B * H * rup(Tq, 256) * rup(d, 16) * 2 + 256; }
size_t tcl_attention_kv_bytes(int Bkv, int H, int Tk, int d) {
    return ((size_t)Bkv * H * rup(Tk, 64) * rup(d, 16) + (size_t)Bkv * H * rup(d, 32) * rup(Tk, 64)) * 2 + 256;
}

// softmax(Q K^T * scale) V per head.  q/k/v point at head 0 of batch 0; row strides ld* and batch strides *bs in halves.
// K/V batch index = b / kv_div (kv_div = F for the text cross-attention whose context repeats per frame, else 1).
// pack_kv = 0 reuses the K/V panels already in ws_kv (same Bkv, H, Tk, d as the call that packed them).
int tcl_attention_f16(const void* q, int ldq, long qbs, const void* k, int ldk, long kbs, const void* v, int ldv, long vbs, void* o, int ldo,
                      long obs, int B, int H, int Tq, int Tk, int d, float scale, int kv_div, int pack_kv, void* ws_q, void* ws_kv,
                      hipStream_t st) {
    TCL_CHECK_ARG(q && o && ws_q && ws_kv && B > 0 && H > 0 && Tq > 0 && Tk > 0 && kv_div > 0 && B % kv_div == 0);
    TCL_CHECK_ARG(d == 40 || d == 80 || d == 160);
    TCL_CHECK_ARG(!pack_kv || (k && v));
    const int Tqp = rup(Tq, 256), Tkp = rup(Tk, 64), DP = rup(d, 16), DPV = rup(d, 32), Bkv = B / kv_div;
    _Float16* Qp = (_Float16*)ws_q;
    _Float16* Kp = (_Float16*)ws_kv;
    _Float16* Vt = Kp + (size_t)Bkv * H * Tkp * DP;
    long qc = (long)B * H * Tqp * (DP / 8), kc = (long)Bkv * H * Tkp * (DP / 8);
    hipLaunchKernelGGL(k_pack_rows, dim3(stream_grid(qc, 256, 2)), dim3(256), 0, st, (const _Float16*)q, qbs, ldq, Tq, H, d,
                       scale * 1.4426950408889634f, Qp, Tqp, DP, qc);
    if (pack_kv) {
        hipLaunchKernelGGL(k_pack_rows, dim3(stream_grid(kc, 256, 2)), dim3(256), 0, st, (const _Float16*)k, kbs, ldk, Tk, H, d, 1.f, Kp, Tkp, DP, kc);
        hipLaunchKernelGGL(k_pack_vt, dim3(Tkp / 64, Bkv * H), dim3(256), (size_t)64 * (DPV + 2) * 2, st, (const _Float16*)v, vbs, ldv, Tk, H, d, Vt, Tkp, DPV);
    }
    static const int nw8 = getenv("TCL_FLASH_NW") ? atoi(getenv("TCL_FLASH_NW")) == 8 : 0;
    const bool big = nw8 && Tq >= 2048;          // 8-wave blocks (256 query rows) halve the K/V traffic per query on long sequences
    if (d == 40) return big ? launch_flash<40, 48, 64, 8>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st)
                            : launch_flash<40, 48, 64, 4>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    if (d == 80) return big ? launch_flash<80, 80, 96, 8>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st)
                            : launch_flash<80, 80, 96, 4>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    return launch_flash<160, 160, 160, 4>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
}

}  // extern "C"
