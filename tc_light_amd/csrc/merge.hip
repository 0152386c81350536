// VidToMe token merging on gfx950 (utils/VidToMe/vidtome/merge.py:20-159 randframe, :343-463 2s; patch.py:14-91).
// The reference materialises scores [2, n_src, n_dst] in f16, cats the batch along dst, takes row max / argsort.
// Here: cosine-normalise rows (f16 semantics), one MFMA kernel computes score tiles and reduces them on the fly to a
// per-src 64-bit key (sortable f16 score << 32 | ~concat_dst_index) with in-lane max + atomicMax -- the score matrix
// never exists -- then a single-block top-r selection on the 16-bit scores picks the merged tokens.  Merging in
// "replace" mode and unmerging are pure row gathers driven by int32 maps built on the device; the global-token bank
// stays on the device (the reference round-trips it through the CPU for every block and chunk, patch.py:65-82).
// Tie rule (the reference's is unspecified on GPU): highest score, then lowest concatenated dst index; equal scores
// keep ascending src order.
#include "common.h"
#include <stdlib.h>
#include "../../include/tclight_hip.h"
#include "prof.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
#define MS 72   // LDS row stride (halves) for a 64-wide K step

// Round 5, second pass: 4 rows per wave with all their loads in flight before the first reduction, NK = ceil(C / 512) chunk slots per lane as a
// template parameter (no branch around a load; chunk / row clamped for the load, masked in the sum and at the store) -- the first form had one
// row per wave and one 16-byte load in flight on 40 of 64 lanes at C = 320.  Per row the same arithmetic and summation order: same bits
// (and the same as elem.hip::k_layernorm<true>, which writes the metric of a block's norm1 output).
template <int NK>
__global__ __launch_bounds__(256) void k_tome_normalize(const _Float16* __restrict__ x, _Float16* __restrict__ y, long rows, int C) {
    constexpr int NR = 4;
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
    if (row0 >= rows) return;
    const int nchunk = C / 8;
    bool ok[NK]; int off[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { const int ch = lane + 64 * k; ok[k] = ch < nchunk; off[k] = min(ch, nchunk - 1) * 8; }
    half8 v[NR][NK];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const _Float16* xr = x + min(row0 + r, rows - 1) * C;
#pragma unroll
        for (int k = 0; k < NK; ++k) v[r][k] = *(const half8*)(xr + off[k]);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const long row = row0 + r;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (ok[k])
#pragma unroll
                for (int j = 0; j < 8; ++j) q += (float)v[r][k][j] * (float)v[r][k][j];
        const float nrm = (float)(_Float16)sqrtf(wave_sum(q));   // norm rounded to f16, then an f16 division
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)((float)v[r][k][j] / nrm);
            if (ok[k] && row < rows) *(half8*)(y + row * C + off[k]) = o;
        }
    }
}

__device__ __forceinline__ unsigned sortable16(_Float16 h) {
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    return (b & 0x8000u) ? (unsigned)(unsigned short)~b : (unsigned)(b | 0x8000u);
}

// Tile epilogue: lane owns src column (lane&31) of each of its 2 column tiles; rows = dst.  Branch-free reduction: packed f16 max
// over the lane's 32 rows, then the lowest row that attains it (descending scan, last write wins) -- the same key as
// max_r (sortable(f16(score)) << 32 | ~index): highest f16 score, ties to the lowest concatenated dst index.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <bool TAIL>
__device__ __forceinline__ void tome_reduce(const float16v (&acc)[2][2], int dj0, int si0, int nb, int na, int bb, int lane,
                                            unsigned long long* __restrict__ keys) {
    const _Float16 ninf = (_Float16)(-65504.f);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        half2v hp[2][8];
        half2v pm = {ninf, ninf};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                half2v h = {(_Float16)acc[a][b][2 * q], (_Float16)acc[a][b][2 * q + 1]};
                if (TAIL) {
                    const int d0 = dj0 + a * 32 + ((2 * q) & 3) + 8 * ((2 * q) >> 2);
                    h[0] = d0 < nb ? h[0] : ninf;
                    h[1] = d0 + 1 < nb ? h[1] : ninf;
                }
                hp[a][q] = h;
                pm = __builtin_elementwise_max(pm, h);
            }
        const _Float16 m = pm[0] > pm[1] ? pm[0] : pm[1];
        int idx = 0;
#pragma unroll
        for (int a = 1; a >= 0; --a)
#pragma unroll
            for (int r = 15; r >= 0; --r) idx = (hp[a][r >> 1][r & 1] == m) ? a * 32 + (r & 3) + 8 * (r >> 2) : idx;
        const int dj = dj0 + idx;
        unsigned long long best = 0ull;
        if (!TAIL || dj < nb) best = ((unsigned long long)sortable16(m) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)(bb * nb + dj));
        unsigned long long other = __shfl_xor(best, 32, 64);
        best = other > best ? other : best;
        if ((lane >> 5) == 0 && si0 + b * 32 < na) atomicMax(keys + si0 + b * 32, best);
    }
}

// keys[src] = max over (batch, dst) of (sortable(f16(score)) << 32 | ~(batch*nb + dst))
// 128 (dst) x 128 (src) score tile per block, K = C in 32-wide steps.  Operand rows are gathered by position (b_pos / a_pos) straight
// into LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB = 16 rows x 64 B per wave instruction; no VGPR staging): 3-stage ring,
// prefetch distance 2, counted vmcnt + one raw barrier per step; rows are unpadded, the 16-B chunk index is XOR-swizzled with
// (row>>2)&3 on the DMA source address and on the ds_read address (conflict-free b128 reads).  48 KiB LDS -> 3 blocks per CU.
// Blocks are XCD-aware: XCD x owns the dst tiles == x (mod 8) and walks the src tiles.
__device__ __attribute__((aligned(16))) unsigned g_tome_zero[64];

__global__ __launch_bounds__(256, 3) void k_tome_match(const _Float16* __restrict__ metric, long bstride, int C, const int* __restrict__ a_pos,
                                                       int na, const int* __restrict__ b_pos, int nb, int tiles_src, int tiles_dst,
                                                       int src_per_blk, unsigned long long* __restrict__ keys) {
    constexpr int STAGE = 256 * 64;                     // bytes: 128 dst rows then 128 src rows, 64 B each
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nrange = (tiles_src + src_per_blk - 1) / src_per_blk;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int srange = j % nrange, tdst = (j / nrange) * 8 + xcd, bb = blockIdx.y;
    if (tdst >= tiles_dst) return;
    const int ts0 = srange * src_per_blk, nts = min(src_per_blk, tiles_src - ts0);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
    const _Float16* base = metric + (long)bb * bstride;
    const _Float16* zero = (const _Float16*)g_tome_zero;
    // DMA lane roles: wave w stages pieces 2w, 2w+1 of each operand; lane -> row rr = lane>>2 of the piece, LDS chunk lane&3
    const int rr = lane >> 2, csrc = ((lane & 3) ^ ((rr >> 2) & 3)) * 8;
    const _Float16* dp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int dj = tdst * 128 + (wid * 2 + i) * 16 + rr;
        dp[i] = dj < nb ? base + (long)b_pos[dj] * C + csrc : nullptr;
    }
    const int nk = C / 32, nstep = nts * nk, frow = lane & 31, fh = lane >> 5;
    // One flattened loop over (src tile, k step): the DMA ring keeps running across tile boundaries, so the prologue of the next src
    // tile and the epilogue of the current one (VALU + atomics) overlap.  The src row pointers of the tile being ISSUED live in sp.
    const _Float16* sp[2];
    int it_issue = 0, kt_issue = 0;
#define TOME_ISSUE(STEP, BUF)                                                                                                 \
    {                                                                                                                         \
        /* steps are issued in order: (tile, k step) of the one being issued are running counters, no division */           \
        const int kt_ = kt_issue;                                                                                             \
        if (kt_ == 0) {                                                                                                       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
                const int si_ = (ts0 + it_issue) * 128 + (wid * 2 + i) * 16 + rr;                                             \
                sp[i] = si_ < na ? base + (long)a_pos[si_] * C + csrc : nullptr;                                              \
            }                                                                                                                 \
        }                                                                                                                     \
        if (++kt_issue == nk) { kt_issue = 0; ++it_issue; }                                                                   \
        char* sb_ = smem + (BUF) * STAGE;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                      \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dp[i] ? dp[i] + kt_ * 32 : zero), \
                                             (__attribute__((address_space(3))) void*)(sb_ + (wid * 2 + i) * 1024), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp[i] ? sp[i] + kt_ * 32 : zero), \
                                             (__attribute__((address_space(3))) void*)(sb_ + 8192 + (wid * 2 + i) * 1024), 16, 0, 0); \
        }                                                                                                                     \
    }
    float16v acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    TOME_ISSUE(0, 0);
    if (nstep > 1) TOME_ISSUE(1, 1);
    int buf = 0, kt = 0, tcur = 0;
    bool drained = false;          // an epilogue's atomics were issued since the last full drain: vmcnt counts are unreliable until vmcnt(0)
    for (int step = 0; step < nstep; ++step) {
        if (step + 1 < nstep && !drained) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // my 4 pieces of this step landed; the next stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        drained = false;
        __builtin_amdgcn_s_barrier();
        if (step + 2 < nstep) { const int nb_ = buf == 0 ? 2 : buf - 1; TOME_ISSUE(step + 2, nb_); }
        const char* db = smem + buf * STAGE + (wm * 64) * 64;
        const char* sbp = smem + buf * STAGE + 8192 + (wn * 64) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { const int R = a * 32 + frow; fa[a] = *(const half8*)(db + R * 64 + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); }
#pragma unroll
            for (int b = 0; b < 2; ++b) { const int R = b * 32 + frow; fb[b] = *(const half8*)(sbp + R * 64 + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        buf = buf == 2 ? 0 : buf + 1;
        if (++kt == nk) {                      // score tile complete: reduce it to keys, start the next src tile
            const int dj0 = tdst * 128 + wm * 64 + 4 * (lane >> 5), si0 = (ts0 + tcur) * 128 + wn * 64 + (lane & 31);
            if (tdst * 128 + 128 > nb) tome_reduce<true>(acc, dj0, si0, nb, na, bb, lane, keys);      // wave-uniform: last dst tile only
            else tome_reduce<false>(acc, dj0, si0, nb, na, bb, lane, keys);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            kt = 0; ++tcur; drained = true;
        }
    }
#undef TOME_ISSUE
}

// ---- C = 320 (level 0: 9/10 of the matching work), src / dst positions affine in the token sequence (every match VidToMe issues: the dst
// frame of a random-frame merge and the src / dst halves of a two-set merge are contiguous runs).
// Same keys as k_tome_match, bit for bit (same MFMA, same K order), different decomposition: the SRC strip is the block's own -- wave w
// keeps the B operand of its 32 src columns in registers for all of K (20 half8 = 80 VGPRs) -- and the block sweeps the dst tiles of its
// range, so the reduction over dst (the expensive part: ~4 vector instructions per score in k_tome_match's tile epilogue, as many
// issue cycles as the tile's MFMAs) becomes a RUNNING packed-f16 maximum in registers: per 32-row MFMA tile 8 cvt_pk + 8 pk_max, and
// the index scan (v_cmp + v_cndmask per element) runs only when some lane's running maximum actually grew -- record-breaking events,
// ~ln(tiles) per column.  No per-tile atomics: one atomicMax per src row and dst split at the end.  LDS holds only dst rows: 64-wide K
// stages (128 B per row, 16 KiB per stage, 4-slot ring), 32 MFMAs per wave and barrier instead of 8; rows are gathered by arithmetic
// (no index loads in the loop, so the counted vmcnt only ever sees the LDS-DMA).  XCD x sweeps dst split x % nsplit: its L2 holds one range.
// LDS image of a stage: row R at R * 128, its 16-B chunk g stored at position g ^ ((R >> 1) & 7): the 16-lane groups of a ds_read_b128
// (MI355X_MICROARCH.md, LDS table) then touch 16 distinct slots of the 256-B bank window.
// Round 6 -- NS = 2, TR = 64: the LDS bound of this kernel.  Every wave of a block reads the SAME dst rows from LDS as the A operand of its MFMAs: 1 KiB per 32x32x16 MFMA, and a CU
// that issues one MFMA per 8 clocks (its four matrix pipes' rate) needs exactly the LDS's 128 B/clk -- the 54 % the matrix pipe showed (profiles/r2_tome_match320_sq_counters.txt) is that
// bound at ~2/3 LDS efficiency, and the flash kernel the chain runs beside lives on the same LDS.  With TWO src sub-tiles per wave (2 x 80 VGPRs of strip) every A fragment feeds two
// MFMAs: half the LDS reads and half the dst DMA per MFMA; the dst tile shrinks to 64 rows so the accumulators stay at 64 VGPRs.  Same MFMA, same K order, same tile order: same keys.
template <int NW, int C = 320, int NS = 1, int TR = 128>        // waves per block: 4 (128-src strip, two blocks per CU) or 8 (256-src strip, one block per CU: half the dst DMA per MFMA); C = 640 (round 4): the level-1 matches, 160 VGPRs of strip
__global__ __launch_bounds__(64 * NW, (NW == 4 && NS == 1) ? 2 : 1) void k_tome_match320(const _Float16* __restrict__ metric, long bstride, int Bt, int a_split, int a_gap, int na,
                                                          int b0, int nb, int tiles_dst, int nsplit, unsigned long long* __restrict__ keys) {
    constexpr int NST = C / 64, STAGE = TR * 128, SW = 32 * NW * NS, NP = (TR / 8) / NW, NA = TR / 32;       // src strip width, DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* const lds0 = (__attribute__((address_space(3))) char*)smem;
    const int bid = blockIdx.x, x = bid & 7, per = 8 / nsplit;
    const int split = x & (nsplit - 1), strip = (bid >> 3) * per + x / nsplit;
    const int tps = (tiles_dst + nsplit - 1) / nsplit, t0 = split * tps, t1 = min(t0 + tps, tiles_dst);
    if (strip * SW >= na || t0 >= t1) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), hl = lane >> 5, col = lane & 31;
    // DMA roles: wave w stages pieces NP w .. NP w + NP - 1 (8 rows x 128 B each); lane -> row rr of the piece, LDS chunk position ch.  A piece's
    // source is a SCALAR base (tile, k stage, piece: s_add / s_addc) plus one of two per-lane byte offsets (the swizzle term (R >> 1) & 7 of
    // row R = 8 piece + rr only depends on the piece's parity): the per-lane 64-bit multiply-add, bounds select and zero page of the
    // first version cost ~8 vector instructions per piece, 31 per K stage next to its 16 MFMAs.  No bounds test at all: the last tile of
    // the sweep is moved back to rows [nb - 128, nb) (needs nb >= 128, host-checked) -- rows seen twice leave a running maximum and its
    // lowest index unchanged.
    const int rr = lane >> 3, ch = lane & 7;
    const int voff0 = rr * (C * 2) + ((ch ^ (rr >> 1)) << 4), voff1 = rr * (C * 2) + ((ch ^ (4 + (rr >> 1))) << 4);
    int si[NS]; long srow[NS];                                                     // this lane's src column(s)
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) { si[s_] = strip * SW + (wid * NS + s_) * 32 + col; srow[s_] = si[s_] < na ? (long)(si[s_] < a_split ? si[s_] : si[s_] + a_gap) * C : -1; }
    const int ntl = t1 - t0, nstep = ntl * NST;
    _Float16 pm[NS]; int bi[NS];                                                    // running maximum of the lane's f16-rounded scores; concatenated dst index where it was first attained
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) { pm[s_] = (_Float16)(-65504.f); bi[s_] = 0x7fffffff; }
    for (int bb = 0; bb < Bt; ++bb) {
        const _Float16* base = metric + (long)bb * bstride;
        half8 bfr[NS][C / 16];                                                      // B operand: src column, k = 16 ks + 8 hl .. + 7
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks) {
                if (srow[s_] >= 0) bfr[s_][ks] = *(const half8*)(base + srow[s_] + ks * 16 + 8 * hl);
                else
#pragma unroll
                    for (int j = 0; j < 8; ++j) bfr[s_][ks][j] = (_Float16)0.f;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the fragment loads are the only non-DMA VMEM reads: drain before counting
        int i_t = 0, i_k = 0;                                                       // (tile, stage) of the step being ISSUED
#define T320_ISSUE(BUF)                                                                                                       \
        {                                                                                                                     \
            const int dj0_ = min((t0 + i_t) * TR, nb - TR);                                                                   \
            const char* sb_ = (const char*)base + ((long)(b0 + dj0_ + wid * (8 * NP)) * C + i_k * 64) * 2;                    \
            _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                                 \
                const char* src_ = sb_ + i * (8 * C * 2) + ((i & 1) ? voff1 : voff0);                                         \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                         \
                                                 (__attribute__((address_space(3))) void*)(lds0 + (BUF) * STAGE + (wid * NP + i) * 1024), 16, 0, 0); \
            }                                                                                                                 \
            if (++i_k == NST) { i_k = 0; ++i_t; }                                                                             \
        }
        T320_ISSUE(0);
        if (nstep > 1) T320_ISSUE(1);
        int step = 0;
        for (int tl = 0; tl < ntl; ++tl) {
            float16v acc[NS][NA];
#pragma unroll
            for (int kt = 0; kt < NST; ++kt, ++step) {
                // 4-slot ring, two K stages per barrier: steps s, s+1 (s even) are consumed while s+2, s+3 stream into the slots of s-2, s-1
                // (without any barrier the kernel ran 4-7 % faster; this halves their number)
                if ((step & 1) == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // my pieces of steps s, s+1 landed (nothing else is in flight)
                    __builtin_amdgcn_s_barrier();
                    if (step + 2 < nstep) T320_ISSUE((step + 2) & 3);
                    if (step + 3 < nstep) T320_ISSUE((step + 3) & 3);
                }
                const char* db = smem + (step & 3) * STAGE;
                // A fragments one k-slice ahead of the MFMAs that use them: the LDS latency of slice ks+1 hides under the 4 MFMAs of slice ks
                // (with the reads issued right before their MFMAs the waves sat parked 45 % of their cycles, matrix pipe 44 % busy)
                constexpr bool PF = C <= 320;                                       // C = 640: the strip takes 160 VGPRs -- no second fragment set (the two waves of a SIMD cover for each other)
                half8 fa[PF ? 2 : 1][NA];
                if (PF) {
#pragma unroll
                    for (int a = 0; a < NA; ++a) { const int R = a * 32 + col; fa[0][a] = *(const half8*)(db + R * 128 + (((0 + hl) ^ ((R >> 1) & 7)) << 4)); }
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (PF ? ks < 3 : true) {
                        const int kn = PF ? ks + 1 : ks;
#pragma unroll
                        for (int a = 0; a < NA; ++a) { const int R = a * 32 + col; fa[PF ? (kn & 1) : 0][a] = *(const half8*)(db + R * 128 + (((2 * kn + hl) ^ ((R >> 1) & 7)) << 4)); }
                    }
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int s_ = 0; s_ < NS; ++s_) {
                            if (kt == 0 && ks == 0) {
                                float16v z;
#pragma unroll
                                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                                acc[s_][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][a], bfr[s_][0], z, 0, 0, 0);
                            } else acc[s_][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[PF ? (ks & 1) : 0][a], bfr[s_][kt * 4 + ks], acc[s_][a], 0, 0, 0);
                        }
                }
            }
            // ---- score tile (128 dst x 32 src per wave) -> running maximum.  f32 -> f16 rounding is monotonic, so the maximum of the 16
            // rounded scores a lane holds of a 32-row tile is the rounded f32 maximum: 8 v_max3 + one conversion instead of 8 cvt_pk + 8
            // pk_max; the conversions of all 16 and the lowest-index scan run only when the running maximum grew (~ln(tiles) times).
            const int dj0 = min((t0 + tl) * TR, nb - TR), cat0 = bb * nb + dj0 + 4 * hl;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                float t = fmaxf(fmaxf(acc[s_][a][0], acc[s_][a][1]), acc[s_][a][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) t = fmaxf(fmaxf(t, acc[s_][a][r]), acc[s_][a][r + 1]);
                t = fmaxf(t, acc[s_][a][15]);
                const _Float16 th = (_Float16)t;
                if (__any(th > pm[s_])) {                                           // some lane's running maximum grew inside this 32-row tile
                    asm volatile("; record");                                       // (a real branch: the scan below is the expensive part)
                    int rlo = 0;                                                    // lowest element attaining the tile maximum (descending scan)
#pragma unroll
                    for (int r = 15; r >= 0; --r) rlo = (_Float16)acc[s_][a][r] == th ? r : rlo;
                    if (th > pm[s_]) { pm[s_] = th; bi[s_] = cat0 + a * 32 + (rlo & 3) + 8 * (rlo >> 2); }
                }
            }
        }
#undef T320_ISSUE
        __builtin_amdgcn_s_barrier();                                               // everyone is done reading the ring before the next batch refills it
    }
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        unsigned long long best = ((unsigned long long)sortable16(pm[s_]) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)bi[s_]);
        const unsigned long long other = __shfl_xor(best, 32, 64);
        best = other > best ? other : best;
        if (hl == 0 && si[s_] < na) atomicMax(keys + si[s_], best);
    }
}

// Top-r selection and map construction without a sort, in TWO launches per match (round 4; rounds 2-3: five -- two histogram passes, two
// single-block picks, the map kernel -- ~650 k launches per 300-frame pass).  The r src tokens with the highest f16 score are merged, ties at the
// threshold go to the lowest src index (= the order a stable descending sort would produce); the reference orders the remaining (unmerged) src
// slots by score as well, but that order is immaterial -- the merged sequence only feeds a permutation-invariant attention and is mapped back by
// `unm` -- so they keep their src index order here.  Maps of SURVEY 8(a) A12/A13:
//  mrg[p]  (p in [0, na-r+nb))  = input position feeding merged slot p          (merge, mode "replace")
//  unm[pos] (pos in input seq)  = merged slot that input position pos is restored from (unmerge)
// k_thr_select: ONE histogram over all 65 536 sortable-f16 scores (global atomics, one per src token: the 256 KiB of bins live at a fixed offset
//   of the workspace and are all-zero between matches); the block that draws the last ticket (guide: "last arriver" hand-off -- writers drain their
//   atomics, release fence, ticket; the last arriver acquires) scans the bins from the top, finds the threshold score and how many of its ties
//   are taken, and zeroes the bins again.  Small blocks (256 threads, no LDS to speak of): in the pipeline this chain runs on a side stream beside
//   the flash kernel, whose two blocks per CU leave only a register / LDS remainder.
// k_tome_maps2: one thread per src / dst token.  A src block needs the number of ties / lower scores BEFORE its first token: it simply counts them
//   over the preceding keys (<= 64 k keys of 8 B, L2-resident: 3.6 M key reads in total at na = 43 200 -- less than one pass of the score kernel's
//   epilogue), in-block prefixes by ballot / popcount.  The last block to finish clears the key array for the next match (last-arriver ticket
//   again), so the workspace is all-zero between matches except the two result words at a fixed offset -- rounds 2-3 left per-chunk scan words
//   behind the keys, which a later match with a LARGER src count then met as initial keys (harmless in practice: they sat below any real
//   threshold; gone now).
#define THR_BS 256
#define THR_CTRL_INTS 8      // ws ints [768, 776): ticket of k_thr_select, ticket of k_tome_maps2, thr, take
__global__ __launch_bounds__(THR_BS) void k_thr_select(const unsigned long long* __restrict__ keys, int na, int r, int* __restrict__ hist16, int* __restrict__ ctrl) {
    __shared__ int sc[THR_BS];
    __shared__ int s_last, s_who;
    const int tid = threadIdx.x;
    for (int i = blockIdx.x * THR_BS + tid; i < na; i += gridDim.x * THR_BS)
        __hip_atomic_fetch_add(hist16 + (int)((keys[i] >> 32) & 0xFFFFu), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(ctrl + 0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_who = -1;
    }
    __syncthreads();
    if (!s_last) return;
    // thread t owns the 256 bins [65535 - 256 t - 255, 65535 - 256 t]: descending score order over t.  Plain 16-B loads behind the acquire (the
    // guide's last-arriver recipe; an agent-scope atomic load per bin made this scan a chain of 256 dependent round trips: 97 us per match in
    // the first profiled pass of the round, more than the five launches it replaced)
    const int hi = 65535 - 256 * tid;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4* bins = (const i32x4*)(hist16 + hi - 255);                   // ascending addresses: bins hi-255 .. hi
    int cnt = 0;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) { const i32x4 v = bins[q]; cnt += v[0] + v[1] + v[2] + v[3]; }
    sc[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < THR_BS; off <<= 1) { const int v = tid >= off ? sc[tid - off] : 0; __syncthreads(); sc[tid] += v; __syncthreads(); }
    const int incl = sc[tid], excl = incl - cnt;
    if (r > 0 && excl < r && r <= incl) s_who = tid;
    __syncthreads();
    if (r <= 0) { if (tid == 0) { ctrl[2] = 0x10000; ctrl[3] = 0; } }        // nothing merged
    else if (tid == s_who) {
        int above = excl, thr = hi;
        for (int q = 0; q < 256; ++q) {
            const int c = hist16[hi - q];
            if (above + c >= r) { thr = hi - q; break; }
            above += c;
        }
        ctrl[2] = thr; ctrl[3] = r - above;                                   // threshold score; ties taken (lowest src index first)
    }
    __syncthreads();
    i32x4* zb = (i32x4*)(hist16 + hi - 255);
    const i32x4 z4 = {0, 0, 0, 0};
#pragma unroll 8
    for (int q = 0; q < 64; ++q) zb[q] = z4;                                  // bins back to zero for the next match
    if (tid == 0) ctrl[0] = 0;
}
// Round 6: the same selection in ONE block without global bins -- a two-pass radix select over the high / low byte of the 16-bit sortable score with 256-bin LDS
// histograms (the keys are <= 64 k x 8 B in L2: read twice).  k_thr_select needs 2-64 blocks resident at once, and beside the flash kernel and the score kernels every
// block of a side-stream launch waits for a CU slot: 60 us per launch in the pass against ~10 alone (profiles/r6_bench_kernel_stats.txt).  One block waits once.
// Same threshold and tie count: thr = the largest score with #(score > thr) < r <= #(score >= thr), take = r - #(score > thr).
__global__ __launch_bounds__(1024) void k_thr_select1(const unsigned long long* __restrict__ keys, int na, int r, int* __restrict__ ctrl) {
    __shared__ int h[256], s_b, s_above;
    const int tid = threadIdx.x;
    if (r <= 0) { if (tid == 0) { ctrl[2] = 0x10000; ctrl[3] = 0; } return; }          // nothing merged (block-uniform)
    int bucket = 0, above = 0;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (tid < 256) h[tid] = 0;
        __syncthreads();
        for (int i = tid; i < na; i += 1024) {
            const int s16 = (int)((keys[i] >> 32) & 0xFFFFu);
            if (pass == 0) atomicAdd(&h[s16 >> 8], 1);
            else if ((s16 >> 8) == bucket) atomicAdd(&h[s16 & 255], 1);
        }
        __syncthreads();
        if (tid < 256) {
            int ab = above;                                                            // keys strictly above this bin (within the running prefix)
            for (int b = 255; b > tid; --b) ab += h[b];
            if (ab < r && r <= ab + h[tid]) { s_b = tid; s_above = ab; }               // exactly one bin satisfies this (r <= the keys counted so far)
        }
        __syncthreads();
        if (pass == 0) { bucket = s_b; above = s_above; }
        else if (tid == 0) { ctrl[2] = (bucket << 8) | s_b; ctrl[3] = r - s_above; }
        __syncthreads();
    }
}
#define TOME_MAPS_BS 256
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(TOME_MAPS_BS) void k_tome_maps2(unsigned long long* __restrict__ keys, int* __restrict__ ctrl, int na, int nb, int r, int nsb,
                                                             const int* __restrict__ a_pos, const int* __restrict__ b_pos, int* __restrict__ mrg,
                                                             int* __restrict__ unm) {
    __shared__ int s_bt[4], s_bl[4], s_ct[4], s_cl[4], s_last;
    const int nun = na - r, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int thr = ctrl[2], take = ctrl[3];
    if ((int)blockIdx.x >= nsb) {
        const int j = ((int)blockIdx.x - nsb) * TOME_MAPS_BS + tid;
        if (j < nb) { const int pos = b_pos[j]; mrg[nun + j] = pos; unm[pos] = nun + j; }
    } else {
        const int i0 = blockIdx.x * TOME_MAPS_BS, i = i0 + tid;
        int tb = 0, lb = 0;                                                   // ties / lower scores among the src tokens before this block
        for (int q = tid; q < i0; q += TOME_MAPS_BS) {
            const int s16 = (int)((keys[q] >> 32) & 0xFFFFu);
            tb += s16 == thr; lb += s16 < thr;
        }
        tb = wave_sum_i(tb); lb = wave_sum_i(lb);
        const unsigned long long k = i < na ? keys[i] : 0ull;
        const int sc = (int)((k >> 32) & 0xFFFFu);
        const unsigned long long mt = __ballot(i < na && sc == thr), ml = __ballot(i < na && sc < thr), below = (1ull << lane) - 1ull;
        if (lane == 0) { s_bt[wid] = tb; s_bl[wid] = lb; s_ct[wid] = __popcll(mt); s_cl[wid] = __popcll(ml); }
        __syncthreads();
        int tie_before = __popcll(mt & below), low_before = __popcll(ml & below);
#pragma unroll
        for (int w = 0; w < 4; ++w) { tie_before += s_bt[w] + (w < wid ? s_ct[w] : 0); low_before += s_bl[w] + (w < wid ? s_cl[w] : 0); }
        if (i < na) {
            const int pos = a_pos[i];
            const bool merged = sc > thr || (sc == thr && tie_before < take);
            if (merged) {
                const unsigned cidx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
                unm[pos] = nun + (int)(cidx % (unsigned)nb);
            } else {
                // unmerged slot = number of unmerged src tokens before i = (#lower before) + (#ties before that were not taken)
                const int slot = low_before + max(tie_before - take, 0);
                mrg[slot] = pos; unm[pos] = slot;
            }
        }
    }
    // every block is done READING keys once its loads have returned; the last one to say so clears them for the next match
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_last = __hip_atomic_fetch_add(ctrl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    for (int q = tid; q < na; q += TOME_MAPS_BS) keys[q] = 0ull;
    if (tid == 0) ctrl[1] = 0;
}
// out[i] = outer[off + inner[i]]  (inner NULL = identity)
__global__ void k_index_compose(const int* __restrict__ outer, const int* __restrict__ inner, int off, int n, int* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = outer[off + (inner ? inner[i] : i)];
}
// out[bb][p] = map[p] >= 0 ? s1[bb][map[p]] : s2[bb][~map[p]]   (map NULL = identity copy of s1)
__global__ void k_gather_rows(const _Float16* __restrict__ s1, long bs1, const _Float16* __restrict__ s2, long bs2, const int* __restrict__ map,
                              _Float16* __restrict__ out, long bso, int n, int C) {
    const int nchunk = C / 8, bb = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n * nchunk; i += (long)gridDim.x * blockDim.x) {
        int p = (int)(i / nchunk), c8 = (int)(i % nchunk) * 8, m = map ? map[p] : p;
        const _Float16* src = m >= 0 ? s1 + bb * bs1 + (long)m * C : s2 + bb * bs2 + (long)(~m) * C;
        *(half8*)(out + bb * bso + (long)p * C + c8) = *(const half8*)(src + c8);
    }
}
// TWO row sets through ONE map (round 6: a token block and its cosine-normalised twin travel together -- the local merge's survivors into their slot of
// the next [src | dst] block, the new bank into the block of the chunk that will meet it; compute_merge keeps no `cat` copy and never normalises twice):
// oa[bb][p] = sa[bb][map[p]], ob[bb][p] = sb[bb][map[p]]  (map NULL = identity).  blockIdx.z picks the pair; 4 rows per trip in flight.
__global__ __launch_bounds__(256) void k_gather_rows_pair(const _Float16* __restrict__ sa, long bsa, const _Float16* __restrict__ sb, long bsb,
                                                          const int* __restrict__ map, _Float16* __restrict__ oa, long boa, _Float16* __restrict__ ob,
                                                          long bob, int n, int C) {
    const int nchunk = C / 8, bb = blockIdx.y;
    const _Float16* src = (blockIdx.z ? sb + bb * bsb : sa + bb * bsa);
    _Float16* out = (blockIdx.z ? ob + bb * bob : oa + bb * boa);
    const long total = (long)n * nchunk, step = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * step < total; i += 4 * step) {
        int p[4], c8[4], m[4];
        half8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long q = i + u * step; p[u] = (int)(q / nchunk); c8[u] = (int)(q % nchunk) * 8; }
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = map ? map[p[u]] : p[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const half8*)(src + (long)m[u] * C + c8[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) *(half8*)(out + (long)p[u] * C + c8[u]) = v[u];
    }
    for (; i < total; i += step) {
        const int p = (int)(i / nchunk), c8 = (int)(i % nchunk) * 8, m = map ? map[p] : p;
        *(half8*)(out + (long)p * C + c8) = *(const half8*)(src + (long)m * C + c8);
    }
}
// h[bb][i] += y[bb][map[i]]   (unmerge + residual)
__global__ void k_gather_add_rows(_Float16* __restrict__ h, long bsh, const _Float16* __restrict__ y, long bsy, const int* __restrict__ map, int n, int C) {
    const int nchunk = C / 8, bb = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n * nchunk; i += (long)gridDim.x * blockDim.x) {
        int p = (int)(i / nchunk), c8 = (int)(i % nchunk) * 8;
        half8 a = *(half8*)(h + bb * bsh + (long)p * C + c8), b = *(const half8*)(y + bb * bsy + (long)(map ? map[p] : p) * C + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (_Float16)((float)a[j] + (float)b[j]);
        *(half8*)(h + bb * bsh + (long)p * C + c8) = a;
    }
}

extern "C" {

int tcl_tome_normalize_f16(const void* x, void* y, long rows, int C, hipStream_t st) {
    TCL_CHECK_ARG(x && y && rows > 0 && C % 8 == 0 && C <= 2048);
    const int nk = (C / 8 + 63) / 64;
    const dim3 grid(cdiv(rows, 16));
    if (nk <= 1) hipLaunchKernelGGL(k_tome_normalize<1>, grid, dim3(256), 0, st, (const _Float16*)x, (_Float16*)y, rows, C);
    else if (nk == 2) hipLaunchKernelGGL(k_tome_normalize<2>, grid, dim3(256), 0, st, (const _Float16*)x, (_Float16*)y, rows, C);
    else if (nk == 3) hipLaunchKernelGGL(k_tome_normalize<3>, grid, dim3(256), 0, st, (const _Float16*)x, (_Float16*)y, rows, C);
    else hipLaunchKernelGGL(k_tome_normalize<4>, grid, dim3(256), 0, st, (const _Float16*)x, (_Float16*)y, rows, C);
    TCL_LAUNCH_RET();
}

size_t tcl_tome_match_workspace_bytes(int na) { return 4096 + 65536 * 4 + ((size_t)na * 8 + 255) / 256 * 256 + 1024; }

// bipartite soft matching (merge.py:84-117 / :389-421 with align_batch): metric [Bt, T, C] normalised rows; src rows a_pos[na],
// dst rows b_pos[nb] (positions in the T sequence, shared by the Bt batch entries); r src tokens get merged.
// Outputs: mrg int32 [na - r + nb], unm int32 [T'] (indexed by input position; every a_pos/b_pos entry is written).
static int g_tome640 = -1;
static int tome_match_impl(const void* metric, long bstride, int Bt, int C, const int* a_pos, int na, const int* b_pos, int nb, int r,
                           int* mrg, int* unm, void* ws, int affine, int a_split, int a_gap, int b0, hipStream_t st) {
    TCL_CHECK_ARG(metric && a_pos && b_pos && mrg && unm && ws && Bt > 0 && na > 0 && nb > 0 && r >= 0 && r <= na && na <= 64 * 1024 && C % 64 == 0);
    TclProfScope ps(TCL_PROF_MATCH, st, 2.0 * na * nb * C * Bt);
    // ws: [4 KiB control: ints 768.. = tickets, thr, take | 65 536 histogram bins | keys na x 8 B]; bins, tickets and keys are all-zero on entry
    // (the caller zeroes the workspace once) and are left all-zero; every region sits at a fixed offset or behind everything else, so a
    // workspace re-used for another na never shows a match a previous match's scratch
    int* ctrl = (int*)ws + 768;
    int* hist16 = (int*)((char*)ws + 4096);
    unsigned long long* keys = (unsigned long long*)((char*)ws + 4096 + 65536 * 4);
    const int ts = cdiv(na, 128), td = cdiv(nb, 128);
    const size_t lds = (size_t)3 * 256 * 64, lds320 = (size_t)4 * 128 * 128;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute((const void*)k_tome_match, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_tome_match320<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds320);
        (void)hipFuncSetAttribute((const void*)k_tome_match320<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds320);
        (void)hipFuncSetAttribute((const void*)(k_tome_match320<4, 640>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds320);
        set = true;
    }
    // tuning / A-B hook: 0 = always the tile-epilogue kernel, 4 / 8 = one src sub-tile per wave (rounds 2-5; 4- / 8-wave blocks), 2 (default since round 6) = two src
    // sub-tiles per wave on 64-row dst tiles.  Round 6, same box (profiles/r6_ab_tome_x2_alone.txt, r6_ab_tome_x2_inpass.txt): ALONE the 2-sub-tile form is 4-27 % SLOWER
    // (276 unified registers: one wave per SIMD where the old form ran two; 43 200 x 14 400: 946 -> 1 070 us) -- and IN THE PASS, where its neighbour on the SIMD is a flash
    // wave either way and the LDS is shared with that kernel, the 60-frame denoise phase is 0.8 % faster in both interleaved pairs (33.26 / 33.31 -> 33.02 / 33.04 s).
    // The pass is what counts.  Same keys bit for bit (test_tome_match_strip_kernel_equals_tile_kernel passes under TCL_TOME320=2 and =4).
    static const int use320 = getenv("TCL_TOME320") ? atoi(getenv("TCL_TOME320")) : 2;
    // C = 640 (level 1) on the strip kernel: built in round 4 (VERDICT r3 #3a), bit-identical, 5-24 % faster per call alone (tools/micro/bench_tome.py:
    // 7 920 x 7 920: 269 -> 205 us) -- and 0.3 % SLOWER in the pass (same-box A/B, profiles/r4_ab_tome640.txt): its strip takes 160 of a wave's 256
    // registers, so a block only starts on a SIMD with 256 free registers, i.e. not beside two flash waves.  Off by default; TCL_TOME640=1 selects it.
    if (g_tome640 < 0) g_tome640 = getenv("TCL_TOME640") ? atoi(getenv("TCL_TOME640")) : 0;
    const int use640 = g_tome640;
    if (affine && (C == 320 || (C == 640 && use640)) && nb >= 128 && use320) {
        // 4-wave blocks by default.  The 8-wave form (256-src strips, one block per CU, half the dst DMA per MFMA) is 3-6 % faster with the GPU
        // to itself from ~13k x 13k tokens up, but in the pipeline the matching chain runs on a side stream beside the flash kernel: a 512-thread
        // block (352 VGPRs per SIMD lane, 64 KiB LDS) only starts on a CU that holds NO flash block, where a 4-wave block shares one with a flash
        // block -- in the profiled pass the 8-wave launches took 2.2 ms on average against 0.9 alone.  use320 = 8 selects it (tools/micro/bench_tome.py).
        const int nw = (use320 == 8 && C == 320) ? 8 : 4;
        const bool x2 = use320 == 2 && C == 320;             // round 6: two src sub-tiles per wave, 64-row dst tiles (half the LDS reads and dst DMA per MFMA)
        const int tsw = cdiv(na, x2 ? 256 : 32 * nw), slots = (nw == 8 || x2) ? 256 : 512, tdx = x2 ? cdiv(nb, 64) : td;
        int nsplit = 1;
        while (nsplit < 8 && (long)tsw * nsplit < slots * 3 / 2) nsplit *= 2;
        while (nsplit > 1 && cdiv(tdx, nsplit) < 2) nsplit /= 2;
        const int per = 8 / nsplit, groups = cdiv(tsw, per);
        if (x2) {
            static bool set2 = false;
            if (!set2) { (void)hipFuncSetAttribute((const void*)(k_tome_match320<4, 320, 2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128); set2 = true; }
            hipLaunchKernelGGL((k_tome_match320<4, 320, 2, 64>), dim3(groups * 8), dim3(256), (size_t)4 * 64 * 128, st, (const _Float16*)metric, bstride, Bt, a_split, a_gap, na, b0, nb, tdx, nsplit, keys);
        } else
        if (C == 640) hipLaunchKernelGGL((k_tome_match320<4, 640>), dim3(groups * 8), dim3(256), lds320, st, (const _Float16*)metric, bstride, Bt, a_split, a_gap, na, b0, nb, td, nsplit, keys);
        else if (nw == 8) hipLaunchKernelGGL(k_tome_match320<8>, dim3(groups * 8), dim3(512), lds320, st, (const _Float16*)metric, bstride, Bt, a_split, a_gap, na, b0, nb, td, nsplit, keys);
        else hipLaunchKernelGGL(k_tome_match320<4>, dim3(groups * 8), dim3(256), lds320, st, (const _Float16*)metric, bstride, Bt, a_split, a_gap, na, b0, nb, td, nsplit, keys);
    } else {
    // each block keeps one dst tile and streams a run of src tiles; runs as long as possible while ~4 blocks per slot (256 CUs x 3) remain
    int spb = (int)((long)ts * td * Bt / 3072);
    if (spb < 1) spb = 1;
    if (spb > 32) spb = 32;
    const int nrange = cdiv(ts, spb);
    hipLaunchKernelGGL(k_tome_match, dim3(cdiv(td, 8) * 8 * nrange, Bt), dim3(256), lds, st, (const _Float16*)metric, bstride, C, a_pos, na, b_pos, nb, ts, td,
                       spb, keys);
    }
    static const int thr1 = getenv("TCL_THR1") ? atoi(getenv("TCL_THR1")) : 1;      // 0 = the multi-block histogram kernel of round 4
    if (thr1) hipLaunchKernelGGL(k_thr_select1, dim3(1), dim3(1024), 0, st, keys, na, r, ctrl);
    else {
        const int hg = na >= 16384 ? 64 : (na >= 2048 ? 16 : 2);
        hipLaunchKernelGGL(k_thr_select, dim3(hg), dim3(THR_BS), 0, st, keys, na, r, hist16, ctrl);
    }
    const int nsb = cdiv(na, TOME_MAPS_BS);
    hipLaunchKernelGGL(k_tome_maps2, dim3(nsb + cdiv(nb, TOME_MAPS_BS)), dim3(TOME_MAPS_BS), 0, st, keys, ctrl, na, nb, r, nsb, a_pos, b_pos, mrg, unm);
    TCL_LAUNCH_RET();
}
int tcl_tome_strip640(int enable) { g_tome640 = enable ? 1 : 0; return TCL_OK; }
int tcl_tome_match_f16(const void* metric, long bstride, int Bt, int C, const int* a_pos, int na, const int* b_pos, int nb, int r,
                       int* mrg, int* unm, void* ws, hipStream_t st) {
    return tome_match_impl(metric, bstride, Bt, C, a_pos, na, b_pos, nb, r, mrg, unm, ws, 0, 0, 0, 0, st);
}
int tcl_tome_match_affine_f16(const void* metric, long bstride, int Bt, int C, const int* a_pos, int na, const int* b_pos, int nb, int r,
                              int a_split, int a_gap, int b0, int* mrg, int* unm, void* ws, hipStream_t st) {
    TCL_CHECK_ARG(a_split >= 0 && a_gap >= 0 && b0 >= 0);
    return tome_match_impl(metric, bstride, Bt, C, a_pos, na, b_pos, nb, r, mrg, unm, ws, 1, a_split, a_gap, b0, st);
}
int tcl_index_compose(const int* outer, const int* inner, int off, int n, int* out, hipStream_t st) {
    TCL_CHECK_ARG(outer && out && n > 0);
    hipLaunchKernelGGL(k_index_compose, dim3(cdiv(n, 256)), dim3(256), 0, st, outer, inner, off, n, out);
    TCL_LAUNCH_RET();
}
int tcl_gather_rows_f16(const void* s1, long bs1, const void* s2, long bs2, const int* map, void* out, long bso, int Bt, int n, int C, hipStream_t st) {
    TCL_CHECK_ARG(s1 && out && Bt > 0 && n > 0 && C % 8 == 0);
    int g = stream_grid((long)n * (C / 8), 256, 2); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_gather_rows, dim3(g, Bt), dim3(256), 0, st, (const _Float16*)s1, bs1, (const _Float16*)s2, bs2, map, (_Float16*)out, bso, n, C);
    TCL_LAUNCH_RET();
}
int tcl_gather_rows_pair_f16(const void* sa, long bsa, const void* sb, long bsb, const int* map, void* oa, long boa, void* ob, long bob, int Bt, int n, int C,
                             hipStream_t st) {
    TCL_CHECK_ARG(sa && oa && (sb == nullptr) == (ob == nullptr) && Bt > 0 && n > 0 && C % 8 == 0);
    int g = stream_grid((long)n * (C / 8), 256, 2); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_gather_rows_pair, dim3(g, Bt, sb ? 2 : 1), dim3(256), 0, st, (const _Float16*)sa, bsa, (const _Float16*)sb, bsb, map, (_Float16*)oa, boa,
                       (_Float16*)ob, bob, n, C);
    TCL_LAUNCH_RET();
}
int tcl_gather_add_rows_f16(void* h, long bsh, const void* y, long bsy, const int* map, int Bt, int n, int C, hipStream_t st) {
    TCL_CHECK_ARG(h && y && Bt > 0 && n > 0 && C % 8 == 0);
    int g = stream_grid((long)n * (C / 8), 256, 2); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_gather_add_rows, dim3(g, Bt), dim3(256), 0, st, (_Float16*)h, bsh, (const _Float16*)y, bsy, map, n, C);
    TCL_LAUNCH_RET();
}

}  // extern "C"
