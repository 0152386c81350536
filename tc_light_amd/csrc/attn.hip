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
#include "gemm_conv.h"
#include "prof.h"
#include "../../include/tclight_hip.h"
#include <stdlib.h>
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef TCL_FLASH80_DEFAULT
#define TCL_FLASH80_DEFAULT 4
#endif
#ifndef TCL_FLASH_PV32_DEFAULT
#define TCL_FLASH_PV32_DEFAULT 0
#endif
#define KV_TILE 64
#define V_STRIDE 72   // halves: 144 B = 9 x 16 B (odd) -> the 16-lane groups of a ds_read_b128 are conflict-free
// Head_dim 40's V^T tile: DPV rows of V_STRIDE halves.  PV16 reads 48 rows (40 + the ones row + 7 idle); rounds 2-4 stored 64 (9 KiB per tile, 16 DMA
// pieces per 64-key stage with the 7 KiB K image).  TCL_DPV40 = 48 (round 5): 48 rows, tile padded to 7 KiB so that a stage is exactly 14 pieces --
// 12.5 % less L2 -> LDS traffic and two DMA instructions less per tile; the PV32 variant needs all 64 rows.
#ifndef TCL_DPV40
#define TCL_DPV40 64
#endif
__host__ __device__ constexpr int vt_tile_halves(int dpv) { return dpv == 48 ? 3584 : dpv * V_STRIDE; }

// one_col >= 0: that column (a padding column, >= d) is set to 1 in valid rows (the K panel's ones column for the folded shift)
__device__ __forceinline__ void pack_rows_blk(int blk, int nblk, const _Float16* __restrict__ src, long bstride, int ld, int T, int H, int d, float scale,
                                              _Float16* __restrict__ dst, int Tp, int DP, long total_chunks, int one_col) {
    const int cpr = DP / 8;
    for (long i = (long)blk * blockDim.x + threadIdx.x; i < total_chunks; i += (long)nblk * blockDim.x) {
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
        if (t < T && one_col >= c8 && one_col < c8 + 8) v[one_col - c8] = (_Float16)1.f;
        *(half8*)(dst + i * 8) = v;
    }
}
__global__ void k_pack_rows(const _Float16* __restrict__ src, long bstride, int ld, int T, int H, int d, float scale,
                            _Float16* __restrict__ dst, int Tp, int DP, long total_chunks, int one_col) {
    pack_rows_blk(blockIdx.x, gridDim.x, src, bstride, ld, T, H, d, scale, dst, Tp, DP, total_chunks, one_col);
}
// Vt panel, tile-major: Vt[bh][tile][i][p(t)] = V[b][tile*64 + t][h*d + i], rows of V_STRIDE halves (64 keys + pad), so that one
// 64-key tile is a contiguous LDS image (DPV x V_STRIDE).  Within every group of 16 keys the two middle blocks of 4 are swapped
// (p swaps bits 2 and 3 of t): the S^T accumulator of the flash kernel leaves lane half hl with keys {4hl..4hl+3, 8+4hl..8+4hl+3}
// of a group, and with this order those 8 V^T values are one contiguous 16-B LDS read (8*hl .. 8*hl+7).  Row D (when DPV > D) is
// a row of ones over the valid keys: the PV MFMA then also yields the softmax row sums.
// skew (head_dim 40, whose PV runs on 16x16x32 MFMAs): rows 4..11 (mod 16) swap the two 16-key halves of every 32-key block (position ^ 16).
// The 16x16x32 A fragment read has lane l fetch row l & 15 at key group l >> 4, and a ds_read_b128 lane group mixes rows 0-3 / 12-15 of one
// key group with rows 4-11 of another: without the skew two pairs of lanes share a 16-B bank slot (SQ_LDS_BANK_CONFLICT = 1/3 of the kernel's
// LDS cycles); with it all 16 slots of the 256-B bank window are distinct (exhaustive check of the four lane groups).
__device__ __forceinline__ void pack_vt_blk(int tile_idx, int bh, const _Float16* __restrict__ v, long bstride, int ld, int T, int H, int d,
                                            _Float16* __restrict__ vt, int ntiles, int DPV, int skew) {
    extern __shared__ _Float16 tile[];   // [64][DPV+2]
    const int t0 = tile_idx * 64, h = bh % H; const long b = bh / H;
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
    _Float16* out = vt + ((long)bh * ntiles + tile_idx) * vt_tile_halves(DPV);
    for (int i = threadIdx.x; i < DPV * 64; i += 256) {
        int dd = i / 64, r = i % 64;
        _Float16 val = tile[r * st + dd];
        if (dd == d && DPV > d) val = (t0 + r < T) ? (_Float16)1.f : (_Float16)0.f;
        int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
        if (skew && (((dd & 15) + 4) & 8)) pr ^= 16;
        out[dd * V_STRIDE + pr] = val;
    }
}
__global__ __launch_bounds__(256) void k_pack_vt(const _Float16* __restrict__ v, long bstride, int ld, int T, int H, int d,
                                                 _Float16* __restrict__ vt, int ntiles, int DPV, int skew) {
    pack_vt_blk(blockIdx.x, blockIdx.y, v, bstride, ld, T, H, d, vt, ntiles, DPV, skew);
}
// Q, K and V^T panels of one attention call in ONE launch (round 4: they were three -- 170 k launches of ~28 us per 300-frame pass on the main
// stream): blocks [0, gq) pack Q rows, [gq, gq + gk) K rows, the rest one V^T tile each.  Same device functions, same bits.
struct PackRows { const _Float16* src; long bstride; int ld, T, d; float scale; _Float16* dst; int Tp, DP; long total; int one_col; };
__global__ __launch_bounds__(256) void k_pack_qkv(PackRows q, PackRows k, int gq, int gk, int H, const _Float16* __restrict__ v, long vbs, int ldv, int Tk, int d,
                                                  _Float16* __restrict__ vt, int ntiles, int DPV, int skew) {
    const int blk = blockIdx.x;
    if (blk < gq) pack_rows_blk(blk, gq, q.src, q.bstride, q.ld, q.T, H, q.d, q.scale, q.dst, q.Tp, q.DP, q.total, q.one_col);
    else if (blk < gq + gk) pack_rows_blk(blk - gq, gk, k.src, k.bstride, k.ld, k.T, H, k.d, k.scale, k.dst, k.Tp, k.DP, k.total, k.one_col);
    else { const int i = blk - gq - gk; pack_vt_blk(i % ntiles, i / ntiles, v, vbs, ldv, Tk, H, d, vt, ntiles, DPV, skew); }
}

// Flash kernel.  Block = 4 waves; each wave owns QB blocks of 32 queries (QB = 2 for head_dim 40: the K and V^T fragments read
// from LDS feed two MFMAs each, which halves the LDS traffic per MFMA -- measured as the co-bottleneck of the one-block version).
// K/V tiles (64 keys) are contiguous "LDS images" in the panels (K rows padded to KS = DP + 8 halves, V^T rows to V_STRIDE), so
// staging is pure LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR staging, no ds_write): ring of NSTG slots,
// prefetch distance NSTG - 1, counted vmcnt + one raw barrier per tile (a __syncthreads() would drain the DMA queue).
// Per tile and query block: S^T = K.Q^T (swapped operands: lane l owns query l&31, so row max/sum are in-lane + one exchange
// with lane l+32), online softmax with lazy rescale (only when some running max grows by more than 2^6), P stays in registers
// as the B operand of O^T += V^T.P^T; row sums come out of the same MFMA through the ones-row of the V^T panel.
// filler source for the DMA pieces behind a stage's end: every lane reads 16 B at lane * 16, so it must span a whole 1 KiB piece (a 256-B
// array let lanes 16-63 read past it -- whatever followed in the code object's data segment, or a fault when that was the segment's end)
__device__ __attribute__((aligned(1024))) unsigned g_flash_zero[256];

// max over the two 32-lane halves without the LDS round trip of a shuffle: v_permlane32_swap exchanges a's upper half with b's lower half
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// (the body of k_flash for ONE block index: the kernel below calls it once, or -- the flag-gated exact pass behind the speculative kernel -- once per
// flagged index of its stride class)
template <int D, int DP, int DPV, int QB, int NSTG, int TPB, int MINB, int SPEC, int PVW, int NW, bool SPLIT = false>
__device__ __forceinline__ void flash_block(const int bid, const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vt,
                                            _Float16* __restrict__ O, int H, int Tq, int Tk, int Tqp, int Tkp, int d, int ldo, long obstride,
                                            int kv_div, int nqb, int* __restrict__ flags, float* __restrict__ lse = nullptr) {
    constexpr int KS = DP + 8;                    // K row stride (halves); KS/8 odd -> conflict-free b128 reads
    constexpr int NQK = DP / 16, NDT = DPV / 32;
    constexpr int KBYTES = KV_TILE * KS * 2, VBYTES = vt_tile_halves(DPV) * 2, SBYTES = KBYTES + VBYTES;
    // NW waves per block (4; 8 for head_dim 80, round 5): a K / V^T stage serves 32 QB NW queries -- its DMA pieces per query halve with 8 waves
    constexpr int NPIECE = (SBYTES + 1023) / 1024, NPW = (NPIECE + NW - 1) / NW, SSTRIDE = NPIECE * 1024;
    constexpr bool LROW = DPV > D;
    constexpr bool FOLD = DP > D && LROW && (D % 16 == 8);        // spare Q/K column D: lanes hl == 1, element 0 of fragment D/16
    constexpr bool PVQ = QB > 1;                                  // PV per query block right behind its softmax (V^T fragments held in registers)
    // PV16 (head_dim 40): O^T = V^T.P^T on 16x16x32 MFMAs over 48 V^T rows (40 + the ones row + 7 idle) instead of 32x32x16 over 64: 12
    // MFMAs of 16 cycles per query block and tile instead of 8 of 32 (-25 % PV matrix time).  The 32x32 S^T accumulator leaves lane l with
    // query l & 31; a 16x16x32 B operand wants query l & 15 in all four 16-lane rows, the rows being four 8-key groups.  One
    // v_permlane16_swap per register pair (first-8-keys register X, second-8-keys register Y of a 32-key block) does exactly that exchange:
    // X.row1 <-> Y.row0, X.row3 <-> Y.row2 turns X into the operand of queries 0-15 and Y into that of queries 16-31, key groups in the order
    // (A, B, C, D) = keys {0-3,8-11}, {16-19,24-27}, {4-7,12-15}, {20-23,28-31} -- which the V^T panel's in-tile key permutation already
    // stores at halves 0, 16, 8, 24 of a 32-key block.
    // PVW = 32 (round 5): head_dim 40 back on 8 MFMAs 32x32x16 over all 64 V^T rows.  The loop is bound by vector ISSUE, not by matrix cycles
    // (tools/micro/flash_mix.hip): the 16 permlane16_swap of a wave-tile and 4 of its 12 + 6 MFMA issues cost more issue slots than the 64
    // matrix cycles PV16 saves; the 32x32 P registers ARE the 32x32x16 B operand (the panel's in-tile key permutation absorbs the key order).
    constexpr bool PV16 = PVW == 16 && D == 40 && LROW && DPV >= 48;
    constexpr int NT16 = 3;                                       // 16-row V^T tiles: rows 0-39 V, row 40 ones, 41-47 zero
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 3 stages of SSTRIDE bytes + 1 KiB dump

    const int head = bid % H, qb_ = (bid / H) % nqb, b = bid / (H * nqb);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), hl = lane >> 5, ql = lane & 31;
    const int q0 = qb_ * (32 * NW * QB) + wid * (32 * QB);
    const long bh = (long)b * H + head, kbh = (long)(b / kv_div) * H + head;
    const int nt = Tkp / KV_TILE, nfull = Tk / KV_TILE;      // tiles / tiles without padded keys
    const char* kbase = (const char*)(Kp + kbh * Tkp * KS);
    const char* vbase = (const char*)(Vt + kbh * nt * vt_tile_halves(DPV));
    const char* zero = (const char*)g_flash_zero;

#ifdef TCL_FLASH_PRIO_EXP
    // lab (round 5, tools/ab/ab_attn.sh): static priority for every other block of an XCD's sequence -- MI355X_MICROARCH "static priority for the
    // younger half", transplanted to two independent 4-wave blocks sharing a CU.  Measured: see DESIGN 4.12 (not in the product build).
    if ((bid >> 3) & 1) __builtin_amdgcn_s_setprio(TCL_FLASH_PRIO_EXP);
#endif
    half8 qf[QB][NQK];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const _Float16* qrow = Qp + (bh * Tqp + q0 + qb * 32 + ql) * DP + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NQK; ++ks) qf[qb][ks] = *(const half8*)(qrow + ks * 16);
    }
    // DMA: piece p = wid + 4 i covers stage bytes [1024 p, 1024 p + 1024); K image first, V^T image behind it.  The piece index is
    // wave-uniform and the K / V^T boundary is piece-aligned (KBYTES = 128 KS, 8 | KS), so a piece's source is a SCALAR base (s_cselect
    // between the two panels) plus the one per-lane offset lane * 16 -- no per-lane selects, no per-piece address registers (an earlier
    // version spent 76 vector instructions per pair of tiles on those selects; a per-piece pointer table spilled the d >= 80 kernels).
    // Only a V^T image that is not a multiple of 1 KiB (DPV = 96) needs the per-lane zero filler behind its end.
    static_assert(KBYTES % 1024 == 0, "K image must be piece-aligned");
    const int lane16 = lane * 16;
    __attribute__((address_space(3))) char* const lds0 = (__attribute__((address_space(3))) char*)smem;
#define FLASH_ISSUE(IT)                                                                                                       \
    {                                                                                                                         \
        const char* kt_ = kbase + (long)(IT) * KBYTES;                                                                        \
        const char* vt_ = vbase + (long)(IT) * VBYTES;                                                                        \
        const int st_ = ((IT) % NSTG) * SSTRIDE;                                                                              \
        _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                                    \
            const int pb_ = (wid + NW * i) * 1024;                                                                            \
            const char* src_ = (pb_ < KBYTES ? kt_ + pb_ : vt_ + (pb_ - KBYTES)) + lane16;                                    \
            if (SBYTES % 1024 != 0) src_ = pb_ + lane16 < SBYTES ? src_ : zero + lane16;                                      \
            else if (NSTG != 3 && NPW * NW != NPIECE && (wid + NW * i) >= NPIECE) continue;   /* nothing to issue past the stage (wave-uniform; the 3-slot ring COUNTS its pieces: it keeps the dump piece) */ \
            const int dst_ = (wid + NW * i) < NPIECE ? st_ + pb_ : NSTG * SSTRIDE;      /* pieces past the stage: 1 KiB dump */  \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                             \
                                             (__attribute__((address_space(3))) void*)(lds0 + dst_), 16, 0, 0);               \
        }                                                                                                                     \
    }

    float16v o[QB][PV16 ? 1 : NDT];
    float4v o16[QB][2][PV16 ? NT16 : 1];          // PV16: [query block][queries 0-15 | 16-31][16-row tile]; lane l: rows 4 (l >> 4) + r, query l & 15
    float m[QB], lsum[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = FOLD ? 0.f : -1e30f; lsum[qb] = 0.f;
#pragma unroll
        for (int t = 0; t < (PV16 ? 1 : NDT); ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][t][r] = 0.f;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int t = 0; t < (PV16 ? NT16 : 1); ++t) o16[qb][qt][t] = float4v{0.f, 0.f, 0.f, 0.f};
    }

    // MK = std::true_type: the tile may hold padded keys (only the last one does).  A compile-time switch, not `if (it >= nfull)`: hipcc turns
    // that runtime test into 126 unconditional v_cmp / v_cndmask / v_add per tile -- half of this VALU-bound loop's vector instructions.
    // SPEC (head_dim 40): the loop is bound by vector ISSUE (tools/micro/flash_mix.hip: its instruction mix allows the matrix pipe 52 % with
    // the row maxima, 62 % without), and the row maximum is the one term that only guards the f16 range of P.  So the shift m is kept OFF = 3
    // bits ABOVE the running row maximum (P <= 2^-3 in the common case; weights below 2^-21 of the row maximum vanish where the textbook P <= 1
    // keeps them down to 2^-24: measured on rows with a sink key 20 bits above the rest, 3.0e-4 rel-L2 against 2.2e-4 for the exact kernel (5.5e-4 with OFF = 4),
    // tests/test_gpu_fullsize.py::test_attention_heavy_tail_and_sink_precision), P is computed without looking at the scores, and the guard reads
    // the result: OR of the packed P registers, bit 14 of a half set <=> some P >= 2 <=> a score rose 4 bits above the maximum the shift was
    // made for.  The guard is evaluated AFTER the tile's PV MFMAs are issued (P in [2, 65504] is still exact, so that PV was right): then
    // `rebase` recomputes the tile's scores from the K tile still in LDS, takes the exact row maxima and moves shift, Q column and O.
    // What this cannot catch in time is a P beyond the f16 range (a score 19 bits above everything the row had seen, inside one tile): the
    // row sum then comes out inf / NaN, the block flags itself and the launch that follows (the exact-maximum kernel, gated by the flags)
    // redoes that block.  Tile 0 starts with a rebase (there is no shift yet).  Scores that keep climbing along the key sequence make every
    // tile pair pay a rebase (keys scaled by a ramp 0.3 .. 6 along the sequence: 567 against 758 TFLOP/s for the exact kernel); keys in
    // token order have no such trend.  Tiles with padded keys (the last one) take the exact-maximum
    // path below -- same shift convention, any shift is valid there -- so the speculative code carries no key masks.
#ifndef TCL_SPEC_OFF
#define TCL_SPEC_OFF 3
#endif
    constexpr float OFF = TCL_SPEC_OFF;
    unsigned orv = 0;                                 // SPEC: OR of the packed P registers since the last guard test
    static_assert(!SPEC || (FOLD && (PV16 || PVQ)), "the speculative softmax is the head_dim-40 path");
    auto rebase = [&](const int it, const bool first) __attribute__((always_inline)) {
        if constexpr (SPEC) {
        const _Float16* kt = (const _Float16*)(smem + (it % NSTG) * SSTRIDE);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float16v s[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NQK; ++ks)
                    s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)(kt + (blk * 32 + ql) * KS + 8 * hl + ks * 16), qf[qb][ks], s[blk], 0, 0, 0);
            }
            float mx = -1e30f;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
            mx = xhalf_max(mx) + OFF;                                            // (scores are relative to the current shift)
            const float mn = (float)(_Float16)(m[qb] + (first ? mx : fmaxf(mx, 0.f)));      // the shift only ever grows; kept f16-representable
            const float delta = mn - m[qb], alpha = __builtin_amdgcn_exp2f(-delta);
            m[qb] = mn;
            if (hl == 1) qf[qb][D / 16][0] = (_Float16)(-mn);                    // Q[q][D] lives in fragment D/16, lanes hl == 1, element 0
            if (!first) {
                if constexpr (PV16) {
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        const float aq = __shfl(alpha, (lane & 15) + 16 * qt, 64);   // this accumulator's query sits in another lane of the S^T layout
#pragma unroll
                        for (int t = 0; t < NT16; ++t) o16[qb][qt][t] *= aq;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NDT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][t][r] *= alpha;           // 32x32 O^T: the query is this lane's own
                }
            }
        }
        }
    };
    auto tile = [&](const int it, auto MK) __attribute__((always_inline)) {
        const _Float16* kt = (const _Float16*)(smem + (it % NSTG) * SSTRIDE);
        const _Float16* vt = kt + KV_TILE * KS;
        constexpr bool mask = decltype(MK)::value;
        half8 kf[2][NQK];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ks = 0; ks < NQK; ++ks) kf[blk][ks] = *(const half8*)(kt + (blk * 32 + ql) * KS + 8 * hl + ks * 16);
        half8 pf[QB][2][2];
        half8 vfr[PVQ && !PV16 ? 2 : 1][2][NDT];     // PVQ: every V^T fragment of the tile in registers (read once, early)
        half8 vf16[PV16 ? 2 : 1][NT16];               // PV16: row t 16 + (l & 15), the 8 keys of group l >> 4 of 32-key block blk
        if constexpr (PV16) {
            const int goff = (((lane >> 4) & 1) * 16 + (lane >> 5) * 8) ^ ((((lane & 15) + 4) & 8) ? 16 : 0);      // (^ 16: the panel's row skew, k_pack_vt)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int t = 0; t < NT16; ++t) vf16[blk][t] = *(const half8*)(vt + (t * 16 + (lane & 15)) * V_STRIDE + blk * 32 + goff);
        } else if constexpr (PVQ) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                    for (int t = 0; t < NDT; ++t) vfr[blk][ss][t] = *(const half8*)(vt + (ql + t * 32) * V_STRIDE + blk * 32 + 16 * ss + 8 * hl);
        }
        // all QK^T MFMAs of the tile first (QB * 2 * NQK back to back): the row-maximum VALU work of query block 0 then runs under the
        // MFMAs of query block 1 still in the pipe, instead of each block's maximum waiting for its own MFMAs with the pipe idle
        float16v sacc[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[qb][blk][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NQK; ++ks) sacc[qb][blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[blk][ks], qf[qb][ks], sacc[qb][blk], 0, 0, 0);
            }
        if constexpr (mask) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { int kv = it * KV_TILE + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl; if (kv >= Tk) sacc[qb][blk][r] = -1e30f; }
        }
        if constexpr (SPEC && !mask) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float16v (&s)[2] = sacc[qb];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pf[qb][blk][r >> 3][r & 7] = (_Float16)__builtin_amdgcn_exp2f(s[blk][r]);
                    u32x4 x = __builtin_bit_cast(u32x4, pf[qb][blk][0]), y = __builtin_bit_cast(u32x4, pf[qb][blk][1]);
                    orv |= x[0] | x[1] | x[2] | x[3] | y[0] | y[1] | y[2] | y[3];
                    if constexpr (PV16) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
                            x[w] = sw[0]; y[w] = sw[1];
                        }
                        const half8 p0 = __builtin_bit_cast(half8, x), p1 = __builtin_bit_cast(half8, y);      // queries 0-15 | 16-31
#pragma unroll
                        for (int t = 0; t < NT16; ++t) {
                            o16[qb][0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf16[blk][t], p0, o16[qb][0][t], 0, 0, 0);
                            o16[qb][1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf16[blk][t], p1, o16[qb][1][t], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                            for (int t = 0; t < NDT; ++t) o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfr[blk][ss][t], pf[qb][blk][ss], o[qb][t], 0, 0, 0);
                    }
                }
            }
            if constexpr (TPB == 1) {
                if (__any((orv & 0x40004000u) != 0u)) {
                    asm volatile("; rebase");                                   // keeps this rare path a real branch
                    rebase(it, false);
                }
                orv = 0;
            }
            return;
        }
        // row maxima of every query block first, then ONE (rare) branch for all re-basing of the tile, so that the common path below --
        // exponentials, conversions and the PV MFMAs of all query blocks -- is a single basic block the scheduler can interleave
        float mxq[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = sacc[qb][0][0];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[qb][blk][r]);
            mxq[qb] = xhalf_max(mx);
        }
        bool need = it == 0;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) need |= FOLD ? (mxq[qb] > 6.f) : (mxq[qb] > m[qb] + 6.f);
        if (__any(need)) {
            asm volatile("; rebase");       // keeps this rare path a real branch (hipcc otherwise runs the multiplies and
                                                         // subtractions below on every tile with alpha = 1 / delta = 0 selected in)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float16v (&s)[2] = sacc[qb];
                const float mx = mxq[qb];
                if constexpr (FOLD) {
                    // The running shift m (kept f16-representable) rides in the spare Q column D against a ones column of the K panel, so
                    // the MFMA already returned s - m and the common path is exp2 alone.  Any shift works as long as every key of the row
                    // uses the same one between rescales; it is re-based when the row maximum climbs more than 2^6 above it (and on tile 0).
                    if (it == 0 || __any(mx > 6.f)) {
                        const float mn = (float)(_Float16)(m[qb] + (it == 0 ? mx : fmaxf(mx, 0.f)));
                        const float delta = mn - m[qb], alpha = __builtin_amdgcn_exp2f(-delta);
                        m[qb] = mn;
                        if (hl == 1) qf[qb][D / 16][0] = (_Float16)(-mn);        // Q[q][D] lives in fragment D/16, lanes hl == 1, element 0
                        if constexpr (PV16) {
#pragma unroll
                            for (int qt = 0; qt < 2; ++qt) {
                                const float aq = __shfl(alpha, (lane & 15) + 16 * qt, 64);      // this accumulator's query sits in another lane of the S^T layout
#pragma unroll
                                for (int t = 0; t < NT16; ++t) o16[qb][qt][t] *= aq;
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < NDT; ++t)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[qb][t][r] *= alpha;
                        }
#pragma unroll
                        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                            for (int r = 0; r < 16; ++r) s[blk][r] -= delta;
                    }
                } else if (__any(mx > m[qb] + 6.f)) {          // lazy rescale (rare after the first tiles)
                    const float mn = fmaxf(m[qb], mx), alpha = __builtin_amdgcn_exp2f(m[qb] - mn);
                    m[qb] = mn;
                    lsum[qb] *= alpha;
#pragma unroll
                    for (int t = 0; t < NDT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][t][r] *= alpha;
                }
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float16v (&s)[2] = sacc[qb];
            float ps = 0.f;
            if (FOLD) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pf[qb][blk][r >> 3][r & 7] = (_Float16)__builtin_amdgcn_exp2f(s[blk][r]);
            } else {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { float p = __builtin_amdgcn_exp2f(s[blk][r] - m[qb]); if (!LROW) ps += p; pf[qb][blk][r >> 3][r & 7] = (_Float16)p; }
            }
            if (!LROW) lsum[qb] += ps;
            if constexpr (PV16) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    u32x4 x = __builtin_bit_cast(u32x4, pf[qb][blk][0]), y = __builtin_bit_cast(u32x4, pf[qb][blk][1]);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
                        x[w] = sw[0]; y[w] = sw[1];
                    }
                    const half8 p0 = __builtin_bit_cast(half8, x), p1 = __builtin_bit_cast(half8, y);      // queries 0-15 | 16-31
#pragma unroll
                    for (int t = 0; t < NT16; ++t) {
                        o16[qb][0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf16[blk][t], p0, o16[qb][0][t], 0, 0, 0);
                        o16[qb][1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf16[blk][t], p1, o16[qb][1][t], 0, 0, 0);
                    }
                }
            } else if constexpr (PVQ) {
                // this query block's PV right behind its softmax: the 4 * NDT MFMAs run while the NEXT block's exponentials issue on the
                // vector pipe (one wave then overlaps its own matrix and vector work instead of leaving that to its SIMD partner)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                        for (int t = 0; t < NDT; ++t) o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfr[blk][ss][t], pf[qb][blk][ss], o[qb][t], 0, 0, 0);
            }
        }
        if constexpr (!PVQ && !PV16) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const _Float16* vr = vt + ql * V_STRIDE + blk * 32 + 16 * ss + 8 * hl;
#pragma unroll
                    for (int t = 0; t < NDT; ++t) {
                        const half8 vf = *(const half8*)(vr + t * 32 * V_STRIDE);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb][blk][ss], o[qb][t], 0, 0, 0);
                    }
                }
        }
    };
    if constexpr (TPB == 2) {
        // 4-slot ring, two tiles per barrier: tiles it, it+1 are consumed while it+2, it+3 stream into the slots of it-2, it-1 --
        // half the rendezvous of the one-tile loop (the waves of a block sit on four SIMDs with different partners and drift apart)
        static_assert(NSTG == 4, "two tiles per barrier need a 4-slot ring");
        FLASH_ISSUE(0);
        if (nt > 1) FLASH_ISSUE(1);
        int it = 0;
        for (; it + 1 < nfull; it += 2) {             // pairs of tiles without padded keys: the hot loop
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // tiles it, it+1 landed for every wave; everyone is done with it-2, it-1
            if (it + 2 < nt) FLASH_ISSUE(it + 2);
            if (it + 3 < nt) FLASH_ISSUE(it + 3);
            if (SPEC && it == 0) rebase(0, true);
            tile(it, std::false_type{});
            tile(it + 1, std::false_type{});
            if constexpr (SPEC) {
                // ONE guard test per pair of tiles, behind both tiles' work (the OR result is long there: no stall on it, and the pair is one
                // basic block); both K tiles are still in their ring slots.  The second rebase sees the first one's shift.
                if (__any((orv & 0x40004000u) != 0u)) {
                    asm volatile("; rebase");                                   // keeps this rare path a real branch
                    rebase(it, false);
                    rebase(it + 1, false);
                }
                orv = 0;
            }
        }
        for (; it < nt; it += 2) {                    // the last one or two tiles (same ring protocol), masked variant
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (it + 2 < nt) FLASH_ISSUE(it + 2);
            if (it + 3 < nt) FLASH_ISSUE(it + 3);
            tile(it, std::true_type{});
            if (it + 1 < nt) tile(it + 1, std::true_type{});
        }
    } else {
        // ring of NSTG slots, prefetch distance NSTG - 1: tile it+NSTG-1 goes into the slot tile it-1 just left
        FLASH_ISSUE(0);
        if (NSTG == 3 && nt > 1) FLASH_ISSUE(1);
        // (two loops over the same ring protocol -- full tiles, then the at most one tile with padded keys -- so that the hot loop holds
        // only the unmasked body: with both bodies in one loop hipcc's hoisted addressing spilled the d >= 80 kernels)
#define FLASH_STEP(MK)                                                                                                        \
        {                                                                                                                     \
            if (NSTG == 3 && it + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");                          \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
            __builtin_amdgcn_s_barrier();             /* every wave's pieces of tile it landed; everyone is done with it-1 */ \
            if (it + NSTG - 1 < nt) FLASH_ISSUE(it + NSTG - 1);                                                               \
            if (SPEC && !decltype(MK)::value && it == 0) rebase(0, true);                                                     \
            tile(it, MK);                                                                                                     \
        }
        int it = 0;
        for (; it < nfull; ++it) FLASH_STEP(std::false_type{})
        for (; it < nt; ++it) FLASH_STEP(std::true_type{})
#undef FLASH_STEP
    }
#undef FLASH_ISSUE
    if constexpr (SPEC) {
        // row sums (O^T row 40) that are not finite and positive: some P left the f16 range -> this block is redone by the gated exact kernel
        bool bad = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if constexpr (PV16) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) { const float l = __shfl(o16[qb][qt][2][0], 32 + (lane & 15), 64); bad |= !(l > 0.f && l < 3e38f); }
            } else { const float l = o[qb][D / 32][4 * ((D % 32) / 8)]; bad |= hl == 0 && !(l > 0.f && l < 3e38f); }     // O^T row D, lanes hl == 0
        }
        const int anybad = __syncthreads_or(bad);
        if (tid == 0) flags[bid] = anybad;
    }
    // ---- epilogue
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        if constexpr (PV16) {
            // O^T row 40 (the ones row: softmax row sums) = tile 2, lanes 32-47, register 0; lane l holds rows 16 t + 4 (l >> 4) + r of query
            // (l & 15) + 16 qt: four consecutive output channels -> one 8-byte store per (qt, t)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float inv = 1.f / __shfl(o16[qb][qt][2][0], 32 + (lane & 15), 64);
                const int q = q0 + qb * 32 + qt * 16 + (lane & 15);
                if (q < Tq) {
                    _Float16* orow = O + (long)b * obstride + (long)q * ldo + head * d;
#pragma unroll
                    for (int t = 0; t < NT16; ++t) {
                        const int dd = t * 16 + 4 * (lane >> 4);
                        if (dd < d) {
                            const float4v v = o16[qb][qt][t];
                            half4 w = {(_Float16)(v[0] * inv), (_Float16)(v[1] * inv), (_Float16)(v[2] * inv), (_Float16)(v[3] * inv)};
                            *(half4*)(orow + dd) = w;
                        }
                    }
                }
            }
            continue;
        }
        float l;
        if (LROW) {      // l sits in O^T row D: tile D/32, register group (D%32)/8 (D%8 == 0), lanes with hl == (D%8)/4 == 0
            constexpr int TL = D / 32, RG = (D % 32) / 8;
            l = o[qb][TL][4 * RG];
            l = __shfl(l, ql, 64);                    // broadcast from the hl == 0 half
        } else {
            l = lsum[qb] + __shfl_xor(lsum[qb], 32, 64);
        }
        const float inv = 1.f / l;
        const int q = q0 + qb * 32 + ql;
        if constexpr (SPLIT) {      // split-KV (k_flash_lse): this entry saw one chunk of the keys -- hand on log2 of its softmax denominator at the shift it used
            if (q < Tq && hl == 0) lse[bh * Tq + q] = m[qb] + __log2f(l);
        }
        if (q < Tq) {
            _Float16* orow = O + (long)b * obstride + (long)q * ldo + head * d;
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int dd = t * 32 + 8 * g + 4 * hl;
                    if (dd < d) {
                        half4 w = {(_Float16)(o[qb][t][4 * g] * inv), (_Float16)(o[qb][t][4 * g + 1] * inv), (_Float16)(o[qb][t][4 * g + 2] * inv), (_Float16)(o[qb][t][4 * g + 3] * inv)};
                        *(half4*)(orow + dd) = w;
                    }
                }
        }
    }
}

template <int D, int DP, int DPV, int QB, int NSTG, int TPB, int MINB, int SPEC, int PVW, int NW>
__global__ __launch_bounds__(64 * NW, (MINB ? MINB : (QB == 1 && DP <= 80 && TPB == 1 ? 3 : 2)) * (NW >= 4 ? NW / 4 : 1)) void k_flash(const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vt,
                                                  _Float16* __restrict__ O, int H, int Tq, int Tk, int Tqp, int Tkp, int d, int ldo, long obstride,
                                                  int kv_div, int nqb, int* __restrict__ flags, int nblk) {
    if constexpr (D == 40 && QB == 2 && !SPEC) {
        // The exact kernel as the second pass behind the speculative one: normally NO block is flagged, and dispatching B H nqb (~3 000) blocks of
        // 222 VGPRs / 66 KiB LDS just to read one word each took 57 us per attention call on the main stream (1.6 s per 300-frame pass).  The
        // gated pass is launched with at most 512 blocks (one resident round); a block walks the flags of its stride class and runs the flagged ones.
        if (flags) {
            // Round 5, second pass: a block reads its flags 64 NW at a time (one load per thread, one round trip per window) instead of one after the other
            // -- a block of a 16-block grid walked ~370 DEPENDENT L2 round trips, which is why small grids measured slower (profiles/r5_ab_flash_gate.txt);
            // with the windowed scan a 32-block grid costs the same as 512 IN THE PASS too (profiles/r5c_ab_gate_inpass.txt: 33.12-33.14 s of denoise at
            // 60 frames either way, 8 blocks +0.3 %): the ~52 us of this launch are not a wait for slots between the matching chain's blocks.  512 stays.
            const int per = (nblk + (int)gridDim.x - 1) / (int)gridDim.x, lo = (int)blockIdx.x * per, hi = min(lo + per, nblk);
            for (int base = lo; base < hi; base += 64 * NW) {
                const int mine = base + (int)threadIdx.x < hi ? flags[base + threadIdx.x] : 0;
                if (!__syncthreads_or(mine)) continue;
                for (int bid = base; bid < min(base + 64 * NW, hi); ++bid) {          // (rare: some block of this window flagged an overflow)
                    if (!flags[bid]) continue;
                    flash_block<D, DP, DPV, QB, NSTG, TPB, MINB, SPEC, PVW, NW>(bid, Qp, Kp, Vt, O, H, Tq, Tk, Tqp, Tkp, d, ldo, obstride, kv_div, nqb, flags);
                    __syncthreads();              // the next item re-uses the LDS ring
                }
            }
            return;
        }
    }
    flash_block<D, DP, DPV, QB, NSTG, TPB, MINB, SPEC, PVW, NW>(blockIdx.x, Qp, Kp, Vt, O, H, Tq, Tk, Tqp, Tkp, d, ldo, obstride, kv_div, nqb, flags);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Head_dim 40, software-pipelined across key tiles (k_flash40p).  The loop of k_flash is bound by VALU issue, and within one wave its
// matrix and vector work are a dependency chain (QK^T -> max -> exp -> PV): the two pipes only overlap where the two waves of a SIMD
// happen to be in opposite phases (matrix pipe 46 %, VALU 68 % busy).  Here every loop iteration i issues three INDEPENDENT streams:
//     vector:  softmax of tile i          (scores from the QK^T MFMAs issued one iteration earlier)
//     matrix:  PV of tile i-1             (P from the softmax of the previous iteration)
//     matrix:  QK^T of tile i+1           (for the softmax of the next iteration)
// so one wave keeps both pipes busy by itself.  The price is two score tiles and two P tiles live: with ONE 32-query block per wave that
// is ~170 VGPRs (two query blocks would need ~300), i.e. each K / V^T fragment read from LDS feeds one MFMA instead of two -- LDS array
// time doubles to ~28 %, still off the critical path.  Same arithmetic as k_flash<40,...> (swapped-operand S^T, shift folded into the
// spare Q column, ones-row row sums, PV on 16x16x32 MFMAs, lazy re-basing when a row maximum climbs 2^6 above the shift).
// Ring of 4 tile slots: iteration i reads V^T of tile i-1 and K of tile i+1 while tile i+2 streams in; one barrier per tile.
// Re-basing at tile i (rare): PV(i-1) is issued early in that path (it belongs to the old shift), O and the scores of tile i move to the
// new shift, and the QK^T of tile i+1 -- issued after the decision -- already sees the new shift in the Q column.
template <int NW, int NSTG>        // waves per block (4: two blocks per CU; 8: one), ring slots (prefetch distance NSTG - 3 tiles beyond the next)
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_flash40p(const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vt,
                                                    _Float16* __restrict__ O, int H, int Tq, int Tk, int Tqp, int Tkp, int ldo, long obstride,
                                                    int kv_div, int nqb) {
    constexpr int D = 40, DP = 48, DPV = 64, KS = DP + 8, NQK = 3, NT16 = 3, PF = NSTG - 3;      // PF: tiles in flight beyond tile it+1
    constexpr int KBYTES = KV_TILE * KS * 2, VBYTES = DPV * V_STRIDE * 2, SBYTES = KBYTES + VBYTES, NPIECE = SBYTES / 1024, NPW = NPIECE / NW, SSTRIDE = SBYTES;
    static_assert(KBYTES % 1024 == 0 && NPIECE % NW == 0 && PF >= 1 && PF <= 3, "tile image must split into NW x NPW pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* const lds0 = (__attribute__((address_space(3))) char*)smem;
    const int bid = blockIdx.x, head = bid % H, qb_ = (bid / H) % nqb, b = bid / (H * nqb);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), hl = lane >> 5, ql = lane & 31;
    const int q0 = qb_ * (NW * 32) + wid * 32;
    const long bh = (long)b * H + head, kbh = (long)(b / kv_div) * H + head;
    const int nt = Tkp / KV_TILE, nfull = Tk / KV_TILE;
    const char* kbase = (const char*)(Kp + kbh * Tkp * KS);
    const char* vbase = (const char*)(Vt + kbh * nt * DPV * V_STRIDE);
    const int lane16 = lane * 16, goff = (((lane >> 4) & 1) * 16 + (lane >> 5) * 8) ^ ((((lane & 15) + 4) & 8) ? 16 : 0);
    half8 qf[NQK];
    {
        const _Float16* qrow = Qp + (bh * Tqp + q0 + ql) * DP + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NQK; ++ks) qf[ks] = *(const half8*)(qrow + ks * 16);
    }
#define P40_ISSUE(IT)                                                                                                         \
    {                                                                                                                         \
        const char* kt_ = kbase + (long)(IT) * KBYTES;                                                                        \
        const char* vt_ = vbase + (long)(IT) * VBYTES;                                                                        \
        const int st_ = ((IT) % NSTG) * SSTRIDE;                                                                              \
        _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                                    \
            const int pb_ = (wid + NW * i) * 1024;                                                                            \
            const char* src_ = (pb_ < KBYTES ? kt_ + pb_ : vt_ + (pb_ - KBYTES)) + lane16;                                    \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                             \
                                             (__attribute__((address_space(3))) void*)(lds0 + st_ + pb_), 16, 0, 0);          \
        }                                                                                                                     \
    }
    float4v o16[2][NT16];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int t = 0; t < NT16; ++t) o16[qt][t] = float4v{0.f, 0.f, 0.f, 0.f};
    float m = 0.f;
    // QK^T of one tile: S^T[key, query] for the wave's 32 queries against the 64 keys of the tile in ring slot (it & 3)
    auto qk = [&](const int it, float16v (&sc)[2]) __attribute__((always_inline)) {
        const _Float16* kt = (const _Float16*)(smem + (it % NSTG) * SSTRIDE);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            half8 kf[NQK];
#pragma unroll
            for (int ks = 0; ks < NQK; ++ks) kf[ks] = *(const half8*)(kt + (blk * 32 + ql) * KS + 8 * hl + ks * 16);
            float16v z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0], z, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < NQK; ++ks) sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], sc[blk], 0, 0, 0);
        }
    };
    // O^T += V^T . P^T of one tile (P already in the 16x16x32 operand layout: [blk][queries 0-15 | 16-31])
    auto pv = [&](const int it, const half8 (&pp)[2][2]) __attribute__((always_inline)) {
        const _Float16* vt = (const _Float16*)(smem + (it % NSTG) * SSTRIDE + KBYTES);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                const half8 vf = *(const half8*)(vt + (t * 16 + (lane & 15)) * V_STRIDE + blk * 32 + goff);
                o16[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pp[blk][0], o16[0][t], 0, 0, 0);
                o16[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pp[blk][1], o16[1][t], 0, 0, 0);
            }
    };
    // exp2 of one score tile -> P in the PV operand layout (cvt_pk pairs + permlane16 exchange, see k_flash PV16)
    auto softmax = [&](const float16v (&sc)[2], half8 (&pp)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            half8 x, y;
#pragma unroll
            for (int r = 0; r < 8; ++r) { x[r] = (_Float16)__builtin_amdgcn_exp2f(sc[blk][r]); y[r] = (_Float16)__builtin_amdgcn_exp2f(sc[blk][8 + r]); }
            u32x4 xu = __builtin_bit_cast(u32x4, x), yu = __builtin_bit_cast(u32x4, y);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const auto sw = __builtin_amdgcn_permlane16_swap(xu[w], yu[w], false, false);
                xu[w] = sw[0]; yu[w] = sw[1];
            }
            pp[blk][0] = __builtin_bit_cast(half8, xu); pp[blk][1] = __builtin_bit_cast(half8, yu);
        }
    };
    // One pipeline step for tile `it`: scores of tile it in sc, P of tile it-1 in pprev; leaves scores of it+1 in snext and P of it in pcur.
    // FIRST: tile 0 (no pending PV, the shift is always based); LAST: no next tile.  The steady state (neither) has NO conditional around its
    // three streams, so they are one basic block for the scheduler; the rare re-basing path carries its own copy of the streams.
    auto step = [&](const int it, float16v (&sc)[2], float16v (&snext)[2], const half8 (&pprev)[2][2], half8 (&pcur)[2][2], auto FIRST, auto LAST)
                    __attribute__((always_inline)) {
        constexpr bool first = decltype(FIRST)::value, last = decltype(LAST)::value;
        if (it >= nfull) {                                                      // wave-uniform: only the last tile has padded keys
            asm volatile("; mask");
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int kv = it * KV_TILE + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl; if (kv >= Tk) sc[blk][r] = -1e30f; }
        }
        float mx = sc[0][0];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[blk][r]);
        mx = xhalf_max(mx);
        if (first || __any(mx > 6.f)) {                                         // re-base the shift (tile 0 always; later only on a 2^6 climb)
            if (!first) { asm volatile("; rebase"); pv(it - 1, pprev); }        // the pending PV belongs to the OLD shift: add it before O is rescaled
            const float mn = (float)(_Float16)(m + (first ? mx : fmaxf(mx, 0.f)));
            const float delta = mn - m, alpha = __builtin_amdgcn_exp2f(-delta);
            m = mn;
            if (hl == 1) qf[D / 16][0] = (_Float16)(-mn);
            if (!first) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const float aq = __shfl(alpha, (lane & 15) + 16 * qt, 64);
#pragma unroll
                    for (int t = 0; t < NT16; ++t) o16[qt][t] *= aq;
                }
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[blk][r] -= delta;
            if (!last) qk(it + 1, snext);
            softmax(sc, pcur);
        } else {
            // the three independent streams of the steady state
            pv(it - 1, pprev);
            if (!last) qk(it + 1, snext);
            softmax(sc, pcur);
        }
    };
    // Before step `it`: tile it+1 (its K is read by this step's QK^T) must have landed -- up to PF younger tiles stay in flight (counted
    // vmcnt); past the barrier every wave has finished step it-1, i.e. the V^T of tile it-2, so slot (it+1+PF) % NSTG is free for tile it+1+PF.
#define P40_SYNC(IT)                                                                                                          \
    if ((IT) + 1 < nt) {                                                                                                      \
        /* in flight here: tiles IT+1 .. IT+PF (those that exist); tile IT+1 has landed once at most the younger ones remain */ \
        if (PF >= 3 && (IT) + 3 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");                          \
        else if (PF >= 2 && (IT) + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");                         \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
        __builtin_amdgcn_s_barrier();                                                                                         \
        if ((IT) + 1 + PF < nt) P40_ISSUE((IT) + 1 + PF);                                                                     \
    }
    float16v sA[2], sB[2];
    half8 pA[2][2], pB[2][2];
    const std::true_type T_{}; const std::false_type F_{};
    P40_ISSUE(0);
#pragma unroll
    for (int j = 1; j <= PF; ++j) if (j < nt) P40_ISSUE(j);                     // tiles 1 .. PF in flight behind tile 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // (prologue: simply everything)
    __builtin_amdgcn_s_barrier();                                               // tile 0 landed for every wave
    qk(0, sA);
    if (nt == 1) { step(0, sA, sB, pB, pA, T_, T_); pv(0, pA); }
    else {
        P40_SYNC(0);
        step(0, sA, sB, pB, pA, T_, F_);                                        // -> P(0) in pA, scores(1) in sB
        int it = 1;
        for (; it + 2 < nt; it += 2) {                                          // it odd; steps it and it+1 are both steady
            P40_SYNC(it);
            step(it, sB, sA, pA, pB, F_, F_);
            P40_SYNC(it + 1);
            step(it + 1, sA, sB, pB, pA, F_, F_);
        }
        if (it + 1 < nt) {                                                      // two tiles left: a steady odd step, then the last (even) one
            P40_SYNC(it);
            step(it, sB, sA, pA, pB, F_, F_);
            step(it + 1, sA, sB, pB, pA, F_, T_);
            pv(it + 1, pA);
        } else {                                                                // one tile left (odd)
            step(it, sB, sA, pA, pB, F_, T_);
            pv(it, pB);
        }
    }
#undef P40_SYNC
#undef P40_ISSUE
    // ---- epilogue (as k_flash PV16): row sums in O^T row 40 = tile 2, lanes 32-47, register 0
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float inv = 1.f / __shfl(o16[qt][2][0], 32 + (lane & 15), 64);
        const int q = q0 + qt * 16 + (lane & 15);
        if (q < Tq) {
            _Float16* orow = O + (long)b * obstride + (long)q * ldo + head * D;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                const int dd = t * 16 + 4 * (lane >> 4);
                if (dd < D) {
                    const float4v v = o16[qt][t];
                    half4 w = {(_Float16)(v[0] * inv), (_Float16)(v[1] * inv), (_Float16)(v[2] * inv), (_Float16)(v[3] * inv)};
                    *(half4*)(orow + dd) = w;
                }
            }
        }
    }
}

// ---- optional in-library timing of the flash kernel (bench.py roofline leg): HIP events recorded on the launch stream
#include <vector>
#include <deque>
struct FlashProf { bool on = false; int dfilter = 0; std::deque<hipEvent_t> ev; double flops = 0.0, ms = 0.0; long launches = 0; int big[4] = {0, 0, 0, 0}; double bigfl = 0.0; };
static FlashProf g_prof;
// resolve (elapsed time -> g_prof.ms) and free the oldest event pairs: all of them (blocking) or only those already complete
static void flash_prof_drain(bool all) {
    while (g_prof.ev.size() >= 2) {
        hipEvent_t e0 = g_prof.ev[0], e1 = g_prof.ev[1];
        if (all) (void)hipEventSynchronize(e1);
        else if (hipEventQuery(e1) != hipSuccess) break;
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        g_prof.ms += t;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        g_prof.ev.pop_front(); g_prof.ev.pop_front();
    }
}

template <int D, int DP, int DPV, int QB, int NSTG, int TPB = 1, int MINB = 0, int SPEC = 0, int PVW = 16, int NW = 4>
static int launch_flash(const _Float16* Qp, const _Float16* Kp, const _Float16* Vt, _Float16* O, int B, int H, int Tq, int Tk, int Tqp, int Tkp,
                        int d, int ldo, long obs, int kv_div, hipStream_t st, int* flags = nullptr, bool count = true) {
    constexpr int SB = KV_TILE * (DP + 8) * 2 + vt_tile_halves(DPV) * 2, NPIECE = (SB + 1023) / 1024;
    const size_t lds = (size_t)NSTG * NPIECE * 1024 + 1024;
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)k_flash<D, DP, DPV, QB, NSTG, TPB, MINB, SPEC, PVW, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    const int nqb = Tqp / (32 * NW * QB);
    const bool prof = g_prof.on && (g_prof.dfilter == 0 || g_prof.dfilter == d);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        if (g_prof.ev.size() > 8192) flash_prof_drain(false);       // a 300-frame pass has ~1e5 launches: keep the live event count bounded
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st);
    }
    const int nblk = B * H * nqb;
    // gated exact pass: a small grid of blocks walks the flags (normally none is set: what the launch costs is the DISPATCH of its 222-VGPR / 66-KiB blocks --
    // 57 us with one block per flag (round 3), 56 us in the pass with one resident round of 512 (round 4).  Round 5 measured smaller grids
    // (TCL_FLASH_GATE_BLOCKS, profiles/r5_ab_flash_gate.txt): 64 blocks the same call rate as 512, 16 blocks 1.4 % SLOWER -- the pass is not dispatch-bound)
    static const int gate_blocks = getenv("TCL_FLASH_GATE_BLOCKS") ? atoi(getenv("TCL_FLASH_GATE_BLOCKS")) : 512;
    const int grid = (D == 40 && QB == 2 && !SPEC && flags && nblk > gate_blocks) ? gate_blocks : nblk;
    hipLaunchKernelGGL((k_flash<D, DP, DPV, QB, NSTG, TPB, MINB, SPEC, PVW, NW>), dim3(grid), dim3(64 * NW), lds, st, Qp, Kp, Vt, O, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, nqb, flags, nblk);
    if (prof) { (void)hipEventRecord(e1, st); g_prof.ev.push_back(e0); g_prof.ev.push_back(e1); if (count) { const double fl = 4.0 * B * H * (double)Tq * Tk * d; g_prof.flops += fl; g_prof.launches++; if (fl > g_prof.bigfl) { g_prof.bigfl = fl; g_prof.big[0] = B; g_prof.big[1] = H; g_prof.big[2] = Tq; g_prof.big[3] = Tk; } } }
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

// ---- round 6: split-KV for the MemFlowNet memory read (head_dim 128, ONE head, ONE entry: 14 400 queries are 114 blocks for 256 CUs -- 512 us per call,
// 0.17 of peak, a quarter of a frame pair).  The keys are cut into `nsplit` chunks that ride as batch entries (the Q panel is packed once per chunk, the
// K / V^T panels of a chunk are a batch entry's), every entry writes its normalised partial output and log2 of its denominator, and k_attn_merge joins them:
// O = sum_s 2^(lse_s - max) O_s / sum_s 2^(lse_s - max).  Same kernel body as k_flash<128, ...> (flash_block with SPLIT).
template <int D, int DP, int DPV, int QB, int NSTG>
__global__ __launch_bounds__(256, 2) void k_flash_lse(const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp, const _Float16* __restrict__ Vt,
                                                      _Float16* __restrict__ O, int H, int Tq, int Tk, int Tqp, int Tkp, int d, int ldo, long obstride, int nqb,
                                                      float* __restrict__ lse) {
    flash_block<D, DP, DPV, QB, NSTG, 1, 0, 0, 16, 4, true>(blockIdx.x, Qp, Kp, Vt, O, H, Tq, Tk, Tqp, Tkp, d, ldo, obstride, 1, nqb, nullptr, lse);
}
// parts [S][Tq][HD] f16, lse [S][H][Tq] -> out [Tq][ldo] (first HD columns)
__global__ void k_attn_merge(const _Float16* __restrict__ parts, const float* __restrict__ lse, _Float16* __restrict__ out, int S, int H, int Tq, int d, int ldo) {
    const int HD = H * d, nchunk = HD / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)Tq * nchunk; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i / nchunk), c8 = (int)(i % nchunk) * 8, h = c8 / d;
        float mx = -3.0e38f;
        for (int sidx = 0; sidx < S; ++sidx) mx = fmaxf(mx, lse[((long)sidx * H + h) * Tq + q]);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
        for (int sidx = 0; sidx < S; ++sidx) {
            const float wgt = exp2f(lse[((long)sidx * H + h) * Tq + q] - mx);
            const half8 pv = *(const half8*)(parts + ((long)sidx * Tq + q) * HD + c8);
            wsum += wgt;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wgt * (float)pv[j];
        }
        const float inv = 1.f / wsum;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)(acc[j] * inv);
        *(half8*)(out + (long)q * ldo + c8) = o;
    }
}

static int launch_flash40p(const _Float16* Qp, const _Float16* Kp, const _Float16* Vt, _Float16* O, int B, int H, int Tq, int Tk, int Tqp, int Tkp,
                           int ldo, long obs, int kv_div, hipStream_t st) {
    constexpr int NW = 8, NSTG = 6;
    const size_t lds = (size_t)NSTG * (KV_TILE * 56 * 2 + 64 * V_STRIDE * 2);
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)k_flash40p<NW, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    const int nqb = Tqp / (NW * 32);
    const bool prof = g_prof.on && (g_prof.dfilter == 0 || g_prof.dfilter == 40);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        if (g_prof.ev.size() > 8192) flash_prof_drain(false);
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st);
    }
    hipLaunchKernelGGL((k_flash40p<NW, NSTG>), dim3(B * H * nqb), dim3(NW * 64), lds, st, Qp, Kp, Vt, O, H, Tq, Tk, Tqp, Tkp, ldo, obs, kv_div, nqb);
    if (prof) { (void)hipEventRecord(e1, st); g_prof.ev.push_back(e0); g_prof.ev.push_back(e1); g_prof.flops += 4.0 * B * H * (double)Tq * Tk * 40; g_prof.launches++; }
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }
// head_dim 40: PV on 32x32x16 MFMAs (1, round 5) or on 16x16x32 MFMAs behind permlane16 swaps (0, rounds 2-4).  The V^T panel's row skew belongs to
// the 16x16x32 fragment read, so the packing and every head_dim-40 kernel variant of a process follow the same switch (TCL_FLASH_PV=16|32).
static int g_pv32 = -1;
static inline bool flash_pv32() {
    if (g_pv32 < 0) { const char* e = getenv("TCL_FLASH_PV"); g_pv32 = e ? (atoi(e) == 32) : TCL_FLASH_PV32_DEFAULT; }
    return g_pv32 != 0;
}

extern "C" {

// Timing of the flash kernel launches (all head dims, or only head_dim == dfilter) with HIP events on their stream.
int tcl_flash_profile_begin(int dfilter) { g_prof.on = true; g_prof.dfilter = dfilter; g_prof.flops = 0.0; g_prof.ms = 0.0; g_prof.launches = 0; g_prof.bigfl = 0.0; g_prof.ev.clear(); return TCL_OK; }
// -> (B, H, Tq, Tk) of the largest launch profiled since begin
int tcl_flash_profile_shape(int* shape4) { TCL_CHECK_ARG(shape4); for (int i = 0; i < 4; ++i) shape4[i] = g_prof.big[i]; return TCL_OK; }
// -> total kernel ms, algorithmic FLOPs (4*B*H*Tq*Tk*d per launch) and launch count since begin; synchronises the events.
int tcl_flash_profile_end(double* total_ms, double* total_flops, long* launches) {
    TCL_CHECK_ARG(total_ms && total_flops && launches);
    flash_prof_drain(true);
    const double ms = g_prof.ms;
    *total_ms = ms; *total_flops = g_prof.flops; *launches = g_prof.launches;
    g_prof.on = false; g_prof.ev.clear();
    return TCL_OK;
}

// panel sizes: Tqp = ceil256(Tq), Tkp = ceil64(Tk), DP = ceil16(d), DPV = ceil32(d); K rows DP+8 halves, V^T tiles DPV x V_STRIDE
size_t tcl_attention_q_bytes(int B, int H, int Tq, int d) { return (size_t)B * H * rup(Tq, 256) * rup(d, 16) * 2 + 256 + (size_t)B * H * (rup(Tq, 256) / 128) * 4; }      // Q panel + per-block flags
size_t tcl_attention_kv_bytes(int Bkv, int H, int Tk, int d) {
    return ((size_t)Bkv * H * rup(Tk, 64) * (rup(d, 16) + 8) + (size_t)Bkv * H * (rup(Tk, 64) / 64) * rup(d, 32) * V_STRIDE) * 2 + 2048;
}

// softmax(Q K^T * scale) V per head.  q/k/v point at head 0 of batch 0; row strides ld* and batch strides *bs in halves.
// K/V batch index = b / kv_div (kv_div = F for the text cross-attention whose context repeats per frame, else 1).
// pack_kv = 0 reuses the K/V panels already in ws_kv (same Bkv, H, Tk, d as the call that packed them).
// The packing half of tcl_attention_f16 on its own (same panels, same workspace layout): Q (scaled) and, with pack_kv, K / V^T.  A caller that
// runs it on another stream than the attention itself passes pack_kv bit 2 (and bit 0 = 0) to tcl_attention_f16 afterwards.
static int attention_pack(const void* q, int ldq, long qbs, const void* k, int ldk, long kbs, const void* v, int ldv, long vbs, int B, int H, int Tq,
                          int Tk, int d, float scale, int kv_div, int pack_q, int pack_kv, void* ws_q, void* ws_kv, hipStream_t st) {
    const int Tqp = rup(Tq, 256), Tkp = rup(Tk, 64), DP = rup(d, 16), KS = DP + 8, DPV = d == 40 && !flash_pv32() ? TCL_DPV40 : rup(d, 32), Bkv = B / kv_div;
    _Float16* Qp = (_Float16*)ws_q;
    _Float16* Kp = (_Float16*)ws_kv;
    _Float16* Vt = Kp + (((size_t)Bkv * H * Tkp * KS + 511) / 512) * 512;        // 1-KiB aligned
    long qc = (long)B * H * Tqp * (DP / 8), kc = (long)Bkv * H * Tkp * (KS / 8);
    if (pack_q && pack_kv) {
        const int gq = stream_grid(qc, 256, 2), gk = stream_grid(kc, 256, 2), nt = Tkp / 64;
        PackRows pq = {(const _Float16*)q, qbs, ldq, Tq, d, scale * 1.4426950408889634f, Qp, Tqp, DP, qc, -1};
        PackRows pk = {(const _Float16*)k, kbs, ldk, Tk, d, 1.f, Kp, Tkp, KS, kc, d == 40 ? d : -1};
        hipLaunchKernelGGL(k_pack_qkv, dim3(gq + gk + nt * Bkv * H), dim3(256), (size_t)64 * (DPV + 2) * 2, st, pq, pk, gq, gk, H, (const _Float16*)v, vbs, ldv, Tk, d, Vt,
                           nt, DPV, d == 40 && !flash_pv32() ? 1 : 0);
        return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
    }
    if (pack_q)
        hipLaunchKernelGGL(k_pack_rows, dim3(stream_grid(qc, 256, 2)), dim3(256), 0, st, (const _Float16*)q, qbs, ldq, Tq, H, d,
                           scale * 1.4426950408889634f, Qp, Tqp, DP, qc, -1);
    if (pack_kv) {
        hipLaunchKernelGGL(k_pack_rows, dim3(stream_grid(kc, 256, 2)), dim3(256), 0, st, (const _Float16*)k, kbs, ldk, Tk, H, d, 1.f, Kp, Tkp, KS, kc,
                           d == 40 ? d : -1);
        hipLaunchKernelGGL(k_pack_vt, dim3(Tkp / 64, Bkv * H), dim3(256), (size_t)64 * (DPV + 2) * 2, st, (const _Float16*)v, vbs, ldv, Tk, H, d, Vt, Tkp / 64, DPV, d == 40 && !flash_pv32() ? 1 : 0);
    }
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}
int tcl_attention_pack_f16(const void* q, int ldq, long qbs, const void* k, int ldk, long kbs, const void* v, int ldv, long vbs, int B, int H, int Tq,
                           int Tk, int d, float scale, int kv_div, int pack_kv, void* ws_q, void* ws_kv, hipStream_t st) {
    TCL_CHECK_ARG(q && ws_q && ws_kv && B > 0 && H > 0 && Tq > 0 && Tk > 0 && kv_div > 0 && B % kv_div == 0);
    TCL_CHECK_ARG(d == 40 || d == 80 || d == 128 || d == 160);
    TCL_CHECK_ARG(!(pack_kv & 1) || (k && v));
    return attention_pack(q, ldq, qbs, k, ldk, kbs, v, ldv, vbs, B, H, Tq, Tk, d, scale, kv_div, 1, pack_kv & 1, ws_q, ws_kv, st);
}
// x: entry b's rows start at x + b * x_bs (elements); row_index (may be NULL): merged token t = row row_index[t] of the entry's block -- the VidToMe merge
// map applied in the GEMM's operand load (merge.py "replace" mode is a pure gather), so the merged sequence needs no tensor of its own.
// attn1's QKV projection of ne x T merged tokens (x [ne*T, K] @ W[3 H d, K]^T, no bias: diffusers Attention.to_q / to_k / to_v) written straight into
// the panels tcl_attention_pack_f16 would have produced from its [ne*T, 3 H d] output -- same bits in every byte the pack writes.  ws_q / ws_kv must have
// been ZERO-INITIALISED once for this (ne, H, T, d) (sizes: tcl_attention_q_bytes / _kv_bytes) and may be reused by stream-ordered calls of the same shape:
// padding rows / columns and the V^T rows above the ones row are never written.  Follow with tcl_attention_f16(..., pack_kv = 4 | pair bit, ws_q, ws_kv).
int tcl_gemm_qkv_panels_f16(const void* x, long x_bs, const int* row_index, const void* W, int ne, int T, int H, int d, int K, int ldx, int ldw, float scale,
                            void* ws_q, void* ws_kv, hipStream_t st) {
    TCL_CHECK_ARG(x && W && ws_q && ws_kv && ne > 0 && T > 0 && H > 0 && (d == 40 || d == 80) && K > 0 && K % 32 == 0 && ldx >= K && ldw >= K);
    const int Tqp = rup(T, 256), Tkp = rup(T, 64), DP = rup(d, 16), KS = DP + 8, DPV = d == 40 && !flash_pv32() ? TCL_DPV40 : rup(d, 32);
    _Float16* Kp = (_Float16*)ws_kv;
    _Float16* Vt = Kp + (((size_t)ne * H * Tkp * KS + 511) / 512) * 512;
    const QkvPanel qp = {(_Float16*)ws_q, Kp, Vt, T, Tqp, Tkp, H, d, DP, KS, DPV, vt_tile_halves(DPV), d == 40 ? 40 : -1, d == 40 && !flash_pv32() ? 1 : 0,
                         scale * 1.4426950408889634f, row_index, x_bs};
    TclProfScope ps(TCL_PROF_GEMM, st, 2.0 * ne * T * 3.0 * H * d * K);
    return gemm_dma_qkv_panels((const _Float16*)x, (const _Float16*)W, ne, K, ldx, ldw, qp, st);
}
int tcl_attention_f16(const void* q, int ldq, long qbs, const void* k, int ldk, long kbs, const void* v, int ldv, long vbs, void* o, int ldo,
                      long obs, int B, int H, int Tq, int Tk, int d, float scale, int kv_div, int pack_kv, void* ws_q, void* ws_kv,
                      hipStream_t st) {
    TCL_CHECK_ARG(q && o && ws_q && ws_kv && B > 0 && H > 0 && Tq > 0 && Tk > 0 && kv_div > 0 && B % kv_div == 0);
    TCL_CHECK_ARG(d == 40 || d == 80 || d == 128 || d == 160);
    const int pair = (pack_kv >> 1) & 1;            // bit 1: these B samples are one half of an identical pair -> pick the kernel variant as for 2 B
    const int prepacked = (pack_kv >> 2) & 1;       // bit 2: tcl_attention_pack_f16 already filled ws_q (and ws_kv): only the attention kernels run
    pack_kv &= 1;
    TCL_CHECK_ARG(!pack_kv || (k && v));
    TCL_CHECK_ARG(!(prepacked && pack_kv));
    const int Tqp = rup(Tq, 256), Tkp = rup(Tk, 64), DP = rup(d, 16), KS = DP + 8, Bkv = B / kv_div;
    _Float16* Qp = (_Float16*)ws_q;
    _Float16* Kp = (_Float16*)ws_kv;
    _Float16* Vt = Kp + (((size_t)Bkv * H * Tkp * KS + 511) / 512) * 512;        // 1-KiB aligned
    if (attention_pack(q, ldq, qbs, k, ldk, kbs, v, ldv, vbs, B, H, Tq, Tk, d, scale, kv_div, !prepacked, pack_kv, ws_q, ws_kv, st) != TCL_OK) return TCL_ELAUNCH;
    // d = 40: two query blocks per wave (shared K/V fragments), 4-slot ring and two tiles per barrier, 2 blocks per CU, when the grid still
    // fills the chip several times over (750 TFLOP/s at T = 35.6k; the variants below reach 700 / 660 / 655 there); else one query block
    // per wave on a 2-slot ring at 4 blocks per CU (107 VGPRs: four waves per SIMD hide each other's softmax; 582 vs 557 TFLOP/s at T = 8.9k
    // for the 3-slot / 3-block variant).  d = 80: one block, 2-slot ring (50 KB LDS -> 3 blocks per CU)
    const bool qb2 = (long)B * (pair ? 2 : 1) * H * (Tqp / 256) >= 1024;
    static const int var40 = getenv("TCL_FLASH40") ? atoi(getenv("TCL_FLASH40")) : 0;      // tuning hook: force a d = 40 variant (tools/ab)
    if (d == 40 && var40 == 4 && TCL_DPV40 == 64 && !flash_pv32()) return launch_flash40p(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, ldo, obs, kv_div, st);
    if (d == 40 && var40 == 2) return launch_flash<40, 48, TCL_DPV40, 1, 3>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    if (d == 40 && var40 == 3) return launch_flash<40, 48, TCL_DPV40, 1, 4, 2, 2>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    if (d == 40 && flash_pv32()) {
        if (qb2 && var40 == 0) {
            int* flags = (int*)((char*)ws_q + (((size_t)B * H * Tqp * DP * 2 + 255) / 256) * 256);
            int rc = launch_flash<40, 48, 64, 2, 4, 2, 0, 1, 32>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st, flags);
            if (rc == TCL_OK) rc = launch_flash<40, 48, 64, 2, 4, 2, 0, 0, 32>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st, flags, false);
            return rc;
        }
        return qb2 && var40 != 1 ? launch_flash<40, 48, 64, 2, 4, 2, 0, 0, 32>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st)
                                 : launch_flash<40, 48, 64, 1, 2, 1, 4, 0, 32>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    }
    if (d == 40 && qb2 && var40 == 0) {
        // speculative softmax (no row maxima in the loop), then the exact kernel over the blocks that flagged an f16 overflow of P (normally none:
        // its blocks read one flag and leave)
        int* flags = (int*)((char*)ws_q + (((size_t)B * H * Tqp * DP * 2 + 255) / 256) * 256);
        int rc = launch_flash<40, 48, TCL_DPV40, 2, 4, 2, 0, 1>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st, flags);
        if (rc == TCL_OK) rc = launch_flash<40, 48, TCL_DPV40, 2, 4, 2>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st, flags, false);
        return rc;
    }
    if (d == 40) return qb2 && var40 != 1 ? launch_flash<40, 48, TCL_DPV40, 2, 4, 2>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st)
                                          : launch_flash<40, 48, TCL_DPV40, 1, 2, 1, 4>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    // d = 80: the 4-wave kernel moves 24.5 KiB of K / V^T image per 128 queries and tile through LDS-DMA -- ~40 B / clk / CU at three blocks per CU, the
    // measured ceiling of that path (profiles/r3_lds_dma_rate.txt); 8 waves per block (TCL_FLASH80=8) halve the bytes per query.
    static const int var80 = getenv("TCL_FLASH80") ? atoi(getenv("TCL_FLASH80")) : TCL_FLASH80_DEFAULT;
    if (d == 80 && var80 == 8 && (long)B * H * (Tqp / 256) >= 256)
        return launch_flash<80, 80, 96, 1, 2, 1, 1, 0, 16, 8>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    if (d == 80) return launch_flash<80, 80, 96, 1, 2>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    if (d == 128) {   // MemFlowNet memory read: ONE head, ONE entry -- at 1280x720 14 400 queries are 114 blocks of 128 for 256 CUs (profiles/r6_memflow_kernel_stats_before.txt:
        // 507 us per call, 0.17 of peak).  Round 6 tried 2-wave blocks (64 queries, twice the blocks; same per-wave work and bits): the frame pair got SLOWER, 47.7
        // against 45.9 ms on one box (profiles/r6_ab_memflow_graph_nw.txt) -- a 2-wave block issues twice the LDS-DMA pieces per wave and hides less of it; the
        // kernel is not simply grid-limited.  TCL_FLASH128_NW=2 selects that form; 4 waves stay the default.
        static const int nw128 = getenv("TCL_FLASH128_NW") ? atoi(getenv("TCL_FLASH128_NW")) : 4;
        if (nw128 == 2)
            return launch_flash<128, 128, 128, 1, 2, 1, 0, 0, 16, 2>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
        return launch_flash<128, 128, 128, 1, 2>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
    }
    return launch_flash<160, 160, 160, 1, 3>(Qp, Kp, Vt, (_Float16*)o, B, H, Tq, Tk, Tqp, Tkp, d, ldo, obs, kv_div, st);
}

// Split-KV attention for one entry (B = 1): see k_flash_lse.  q [Tq, ldq], k [Tk, ldk], v [Tk, ldv] (head h at column h d), o [Tq, ldo].
// Tk must be a multiple of 64 nsplit; d = 128.  ws: tcl_attention_splitkv_workspace_bytes(nsplit, H, Tq, Tk, d).
size_t tcl_attention_splitkv_workspace_bytes(int nsplit, int H, int Tq, int Tk, int d) {
    const size_t qb = (tcl_attention_q_bytes(nsplit, H, Tq, d) + 1023) / 1024 * 1024, kb = (tcl_attention_kv_bytes(nsplit, H, Tk / (nsplit > 0 ? nsplit : 1), d) + 1023) / 1024 * 1024;
    return qb + kb + (((size_t)nsplit * Tq * H * d * 2 + 1023) / 1024 * 1024) + (size_t)nsplit * H * Tq * 4 + 1024;
}
int tcl_attention_splitkv_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int H, int Tq, int Tk, int d,
                              float scale, int nsplit, void* ws, hipStream_t st) {
    TCL_CHECK_ARG(q && k && v && o && ws && H > 0 && Tq > 0 && Tk > 0 && d == 128 && nsplit >= 2 && nsplit <= 16 && Tk % (64 * nsplit) == 0 && ldo >= H * d);
    const int chunk = Tk / nsplit, Tqp = rup(Tq, 256), Tkp = chunk, DP = 128, KS = DP + 8;
    char* base = (char*)ws;
    const size_t qb = (tcl_attention_q_bytes(nsplit, H, Tq, d) + 1023) / 1024 * 1024, kb = (tcl_attention_kv_bytes(nsplit, H, chunk, d) + 1023) / 1024 * 1024;
    void *ws_q = base, *ws_kv = base + qb;
    _Float16* parts = (_Float16*)(base + qb + kb);
    float* lse = (float*)(base + qb + kb + (((size_t)nsplit * Tq * H * d * 2 + 1023) / 1024 * 1024));
    if (attention_pack(q, ldq, 0, k, ldk, (long)chunk * ldk, v, ldv, (long)chunk * ldv, nsplit, H, Tq, chunk, d, scale, 1, 1, 1, ws_q, ws_kv, st) != TCL_OK) return TCL_ELAUNCH;
    _Float16* Qp = (_Float16*)ws_q;
    _Float16* Kp = (_Float16*)ws_kv;
    _Float16* Vt = Kp + (((size_t)nsplit * H * Tkp * KS + 511) / 512) * 512;
    constexpr int NSTG = 2, SB = KV_TILE * (128 + 8) * 2 + vt_tile_halves(128) * 2, NPIECE = (SB + 1023) / 1024;
    const size_t lds = (size_t)NSTG * NPIECE * 1024 + 1024;
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)k_flash_lse<128, 128, 128, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
    const int nqb = Tqp / 128;
    hipLaunchKernelGGL((k_flash_lse<128, 128, 128, 1, 2>), dim3(nsplit * H * nqb), dim3(256), lds, st, Qp, Kp, Vt, parts, H, Tq, chunk, Tqp, Tkp, d, H * d,
                       (long)Tq * H * d, nqb, lse);
    hipLaunchKernelGGL(k_attn_merge, dim3(stream_grid((long)Tq * (H * d / 8), 256, 2)), dim3(256), 0, st, parts, lse, (_Float16*)o, nsplit, H, Tq, d, ldo);
    TCL_LAUNCH_RET();
}

}  // extern "C"
