// 8-phase 256-row fp16 MFMA GEMM / implicit 3x3 convolution for gfx950 (round 4): the large-problem path behind tcl_gemm_f16 /
// tcl_conv3x3_f16 (same contract as gemm.hip: C = act(A.W^T + bias) + resid, f32 accumulate; reference call sites
// generate.py:342-347 -> diffusers UNet2DConditionModel convs / Linears).
//
// Structure = the "256^2 8-phase" template of /opt/skills/guides/cdna_hip_programming.md:612-660, measured in tools/micro/gemm8ph_lab.hip
// (profiles/r4_gemm8ph_lab.txt: 1.27-1.32 PFLOP/s where k_gemm8p reaches 0.83-1.08 on the same random operands), generalised to the UNet's
// 320-multiple channel counts:
//   * block = 8 waves (two per SIMD, one block per CU), tile 256 x BN, K tile 64 (128-B LDS rows: every LDS-DMA instruction moves 8 whole lines);
//   * wave (wr, wc) of a WM x WN grid owns the four QM x QN "quadrants" (h, g) of its 2 QM x 2 QN outputs; the A (B) tile of a K step lives in LDS
//     as two half-tiles A0 / A1 (B0 / B1) holding the quadrant-h (g) rows of ALL waves, so every wave reads the same half-tile in the same phase
//     and a half-tile is dead as soon as that phase is over -- that is what lets a 2-buffer ring keep three stages in flight;
//   * 4 phases per K tile (8 per loop iteration = two tiles in the two buffers), each {ds_read one or two sub-tiles, issue the LDS-DMA of ONE
//     half-tile, barrier, lgkmcnt(0), 16-20 x mfma_f32_16x16x32_f16 of one quadrant, barrier}; the wave groups wr-half 0 / 1 (= the two waves of
//     every SIMD) run one barrier apart, so one feeds the matrix pipe while the other reads / issues;
//   * with P = the operand whose sub-tile is smaller (kept in registers for both halves) and S = the other one:
//       phase 1: read P0 (first; retired by a counted lgkmcnt BEFORE the barrier), S0   MFMA (P0,S0)   stage S1(t+1)
//       phase 2: read P1                                                                 MFMA (P1,S0)   stage P0(t+2)   [dead since phase 1]
//       phase 3: read S1 (into S0's registers)                                           MFMA (P1,S1)   stage S0(t+2)   [dead since phase 1]
//       phase 4: --                                                                      MFMA (P0,S1)   stage P1(t+2)   [dead since phase 2]; vmcnt(NW)
//     NW = loads of the three youngest stages (P0, S0, P1 of tile t+2): everything older -- all of tile t+1 -- has landed and is read from the next
//     phase on, i.e. behind the barrier that follows every wave's wait.  vmcnt is never 0 inside the loop.
//   RAW / WAR in barrier intervals ("ticks"; group 0 runs phase p's read part in tick 2(p-1), its MFMA part in tick 2(p-1)+1, group 1 one tick later):
//     P0(t): last read G1 tick 1, retired before the barrier ending tick 1 (counted lgkmcnt) -> restaged by G0 in tick 2.
//     S0(t): last read G1 tick 1, retired (lgkmcnt 0) in tick 2 -> restaged G0 tick 4.   P1(t): G1 tick 3 -> G0 tick 6.   S1(t): G1 tick 5 -> G0 tick 8.
//     G1's phase-4 wait sits before the barrier ending tick 7; the first read of tile t+1 is G0's in tick 8.
//   * configurations (+ 512 x 128, WM 8 x WN 1, quadrant 32 x 64, for the VAE's 128-channel layers): 256 x 256 (WM 2, WN 4: quadrant 64 x 32, P = B: the guide's geometry; also the GEGLU feed-forwards -- a wave's 64 columns
//     are one [32 value | 32 gate] group, combined in registers) and 256 x 320 (WM 4, WN 2: quadrant 32 x 80, P = A: every UNet width is a
//     multiple of 320);
//   * LDS image: row R of a half-tile <-> logical row (R / Q) * 2Q + h * Q + R % Q (Q = QM | QN), 16-B chunk index XOR (R >> 1) & 7 applied on the
//     DMA SOURCE address and on the ds_read address (conflict-free b128 reads of 16 consecutive 128-B rows; the DMA destination is lane-linear);
//   * addressing as in k_gemm8p: `buffer_load_dwordx4 ... offen lds` through one descriptor per operand, a 32-bit per-lane offset fixed for the
//     whole K loop, the K step as a scalar; conv: per staged row the base offset of its output pixel and a 9-bit tap mask, invalid taps / rows
//     read offset 0xffffffff (outside the descriptor: the hardware returns zeros); K walks channel-slice-major (conv_kmap, 64-channel slices = one K tile);
//   * the MFMA takes the WEIGHT fragment as its row operand: a lane then holds 4 consecutive output columns of one row; the epilogue packs them to
//     8 B, trades halves between two row blocks with v_permlane16_swap (-> 8 consecutive columns of one row per lane) and writes 16-B row chunks
//     straight from registers (+ residual / GEGLU) -- same rounding points as every other configuration of tcl_gemm_f16 (gemm.hip), so results
//     stay bit-identical across configurations.
#include "common.h"
#include "../../include/tclight_hip.h"
#include "gemm_conv.h"
#include <stdlib.h>

typedef _Float16 q_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 q_half4 __attribute__((ext_vector_type(4)));
typedef float q_float4 __attribute__((ext_vector_type(4)));

#define Q_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#ifndef G8Q_ABL
#define G8Q_ABL 0      // lab-only knock-outs: 1 = epilogue without its global stores, 2 = no epilogue at all, 4 = no K loop (results are garbage)
#endif

template <int WM, int WN, int RI, int CJ, bool CONV, bool UPS = false>
__global__ __launch_bounds__(512) void k_gemm8q(const _Float16* __restrict__ A, const _Float16* __restrict__ W, const _Float16* __restrict__ bias,
                                                const _Float16* __restrict__ resid, _Float16* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                int ldc, int ldr, int act, ConvP cp, int tiles_m, int tiles_n, unsigned long a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WM * WN == 8, "8 waves");
    constexpr int QM = RI * 16, QN = CJ * 16, BM = 2 * WM * QM, BN = 2 * WN * QN;
    static_assert(BM == 256 || BM == 512, "256- / 512-row tiles");
    constexpr bool PB = CJ <= RI;                               // the persistent operand is B (else A)
    constexpr int HB = WN * QN;                                 // rows of a B half-tile
    constexpr int A_HALF = WM * QM * 128, B_HALF = HB * 128, BUF = 2 * A_HALF + 2 * B_HALF;
    constexpr int NA = A_HALF / 8192, NB = (B_HALF + 8191) / 8192;      // LDS-DMA instructions per wave and half-tile
    static_assert(A_HALF % 8192 == 0, "whole A passes");
    constexpr int NRP = PB ? 2 * CJ : 2 * RI, NRS = PB ? 2 * RI : 2 * CJ;      // ds_read_b128 per persistent / streamed sub-tile
    constexpr bool RR = !PB;                                    // re-read P0 in phase 4 instead of keeping both P halves in registers
    constexpr int NP_ = PB ? NB : NA, NS_ = PB ? NA : NB;       // loads per stage of the persistent / streamed operand
    constexpr int NW = RR ? (NS_ + NP_ + NS_) : (NP_ + NS_ + NP_);      // loads of the three youngest stages
    constexpr int DUMP = 2 * BUF;                               // 4 KiB behind the ring: dummy pieces of a partly used B pass
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;       // bijective XCD remap: XCD x works on a contiguous run of row tiles
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;                     // wr-half 0 = waves 0-3, 1 = waves 4-7 (the two waves of a SIMD)
    const int grp = wid >> 2;

    // A is addressed RELATIVE to the block's first row (dense) / first image (conv): the 32-bit per-lane offsets then span one tile (<= 3 images),
    // whatever the size of the whole operand (block-major passes carry > 4 GiB activations: 3.5 M rows x 1280 channels)
    const int b0_ = CONV ? m0 / (cp.Hout * cp.Wout) : 0;
    const unsigned long a_base = CONV ? (unsigned long)b0_ * cp.Hin * cp.Win * cp.Cin * 2ul : (unsigned long)m0 * (unsigned long)lda * 2ul;
    const unsigned long a_rem = a_bytes - a_base;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)A + a_base), 0, (unsigned)(a_rem > 0xffffff00ul ? 0xffffff00ul : a_rem), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);
    // ---- staging: thread <-> (LDS row R = 64 p + (tid >> 3), chunk' = tid & 7) of a half-tile; source chunk = chunk' ^ ((R >> 1) & 7)
    const int srow = tid >> 3;
    const unsigned csrc = (unsigned)(((tid & 7) ^ ((srow >> 1) & 7)) * 16);
    unsigned aoff[NA][2]; unsigned amask[NA];                    // [pass][half]; conv: per pass 2 x 9 tap bits
    unsigned aw[UPS ? NA : 1][2];                                // UPS: per (pass, half) 9 tap bits | 3 x 2-bit source-row deltas | 3 x 2-bit source-column deltas
#pragma unroll
    for (int p = 0; p < NA; ++p) amask[p] = 0u;
#pragma unroll
    for (int p = 0; p < NA; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int R = 64 * p + srow, m = m0 + (R / QM) * (2 * QM) + h * QM + R % QM;
            if (!CONV) aoff[p][h] = (unsigned)(min(m, M - 1) - m0) * (unsigned)lda * 2u + csrc;
            else if (!UPS) {
                const int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
                const int iy0 = oy * cp.stride - cp.pad, ix0 = ox * cp.stride - cp.pad;
                aoff[p][h] = (unsigned)((((b - b0_) * cp.Hin + iy0) * cp.Win + ix0) * cp.Cin) * 2u + csrc;      // wraps for border pixels; only used with a valid tap
                unsigned mk = 0;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int iy = iy0 + tap / 3, ix = ix0 + tap % 3;
                    if (m < M && iy >= 0 && iy < cp.Hin && ix >= 0 && ix < cp.Win) mk |= 1u << tap;
                }
                amask[p] |= mk << (9 * h);
            } else {
                // nearest up-sampling fused in the gather (stride 1, pad 1; UNet up-samplers, scale Hin / Hup in (0.5, 1]): the three logical rows
                // oy - 1 .. oy + 1 of the up-sampled image map to source rows base, base + {0,1}, base + {0,1,2} (k_gemm8s's own formula:
                // min(floor(y * sy), Hin - 1)); the per-lane word keeps those deltas, the tap picks one (2 bit-field extracts + 2 mads per piece)
                const int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
                auto sy_ = [&](int y) { return min((int)floorf(min(max(y, 0), cp.Hup - 1) * cp.sy), cp.Hin - 1); };
                auto sx_ = [&](int x) { return min((int)floorf(min(max(x, 0), cp.Wup - 1) * cp.sx), cp.Win - 1); };
                const int by = sy_(oy - 1), bx = sx_(ox - 1);
                aoff[p][h] = (unsigned)((((b - b0_) * cp.Hin + by) * cp.Win + bx) * cp.Cin) * 2u + csrc;
                unsigned wd = 0;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int iy = oy - 1 + tap / 3, ix = ox - 1 + tap % 3;
                    if (m < M && iy >= 0 && iy < cp.Hup && ix >= 0 && ix < cp.Wup) wd |= 1u << tap;
                }
#pragma unroll
                for (int d = 0; d < 3; ++d) { wd |= (unsigned)(sy_(oy - 1 + d) - by) << (9 + 2 * d); wd |= (unsigned)(sx_(ox - 1 + d) - bx) << (15 + 2 * d); }
                aw[p][h] = wd;
            }
        }
    unsigned woff[NB];                                          // [pass]; half g adds QN rows (scalar)
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int R = min(64 * p + srow, HB - 1), n = n0 + (R / QN) * (2 * QN) + R % QN;
        woff[p] = (unsigned)min(n, N - QN - 1) * (unsigned)ldw * 2u + csrc;
    }
    const unsigned w_g = (unsigned)QN * (unsigned)ldw * 2u;
    char* const sdst = smem + wid * 1024;
    // ---- fragment reads: lane -> row (lane & 15) of a 16-row block, 16-B chunk (kk * 4 + (lane >> 4)) ^ ((row >> 1) & 7)
    const int frow = lane & 15, fc = lane >> 4, fs = (frow >> 1) & 7;
    const int foff0 = frow * 128 + ((fc ^ fs) << 4), foff1 = frow * 128 + (((4 + fc) ^ fs) << 4);
    const char* const a_rd = smem + wr * QM * 128;              // + buffer + half h * A_HALF + i * 2048
    const char* const b_rd = smem + 2 * A_HALF + wc * QN * 128; // + buffer + half g * B_HALF + j * 2048

    q_float4 acc[2][2][RI][CJ];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < CJ; ++j) acc[h][g][i][j] = q_float4{0.f, 0.f, 0.f, 0.f};
    q_half8 fa[1][RI][2], fb[2][CJ][2];                         // A: one register set (P = B: streamed; P = A: re-read); B: both halves when P = B
    const int nt = K / 64;

    // scalars of the K tile being staged (conv: tap + channel slice -> input offset / weight column); SC_(U) is called once per tile
    unsigned s_ka = 0, s_kw = 0; int s_tap = 0, s_shy = 9, s_shx = 15;
    const unsigned c2_ = (unsigned)cp.Cin * 2u, wc2_ = (unsigned)cp.Win * c2_;
#define Q_SCAL(U)                                                                                                             \
    {                                                                                                                         \
        int ka_ = (U) * 64, kw_ = ka_; s_tap = 0;                                                                             \
        if (CONV) {                                                                                                           \
            int c0_; s_tap = conv_kmap(ka_, cp.Cin, c0_); kw_ = s_tap * cp.Cin + c0_;                                         \
            if (UPS) { ka_ = c0_; s_shy = 9 + 2 * (s_tap / 3); s_shx = 15 + 2 * (s_tap % 3); }                                \
            else ka_ = ((s_tap / 3) * cp.Win + s_tap % 3) * cp.Cin + c0_;                                                     \
        }                                                                                                                     \
        s_ka = (unsigned)ka_ * 2u; s_kw = (unsigned)kw_ * 2u;                                                                 \
    }
    // one half-tile of tile U (scalars already set) into buffer U & 1
#define Q_STAGE_A(H, U)                                                                                                       \
    {                                                                                                                         \
        char* d_ = sdst + ((U) & 1) * BUF + (H) * A_HALF;                                                                     \
        _Pragma("unroll") for (int p_ = 0; p_ < NA; ++p_) {                                                                  \
            if (!CONV) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, Q_LDS(d_ + p_ * 8192), 16, aoff[p_][H], s_ka, 0, 0);      \
            else if (!UPS) {                                                                                                  \
                const unsigned vo_ = ((amask[p_] >> (9 * (H) + s_tap)) & 1u) ? aoff[p_][H] + s_ka : 0xffffffffu;       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, Q_LDS(d_ + p_ * 8192), 16, vo_, 0, 0, 0);                        \
            } else {                                                                                                          \
                const unsigned w_ = aw[UPS ? p_ : 0][H];                                                                      \
                const unsigned o_ = aoff[p_][H] + ((w_ >> s_shy) & 3u) * wc2_ + ((w_ >> s_shx) & 3u) * c2_ + s_ka;            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, Q_LDS(d_ + p_ * 8192), 16, ((w_ >> s_tap) & 1u) ? o_ : 0xffffffffu, 0, 0, 0); \
            }                                                                                                                 \
        }                                                                                                                     \
    }
#define Q_STAGE_B(G, U)                                                                                                       \
    {                                                                                                                         \
        char* d_ = sdst + ((U) & 1) * BUF + 2 * A_HALF + (G) * B_HALF;                                                        \
        _Pragma("unroll") for (int p_ = 0; p_ < NB; ++p_) {                                                                  \
            char* dd_ = (p_ * 8192 + 8192 <= B_HALF || wid * 1024 + p_ * 8192 < B_HALF) ? d_ + p_ * 8192 : smem + DUMP + (wid & 3) * 1024; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, Q_LDS(dd_), 16, woff[p_], s_kw + (G) * w_g, 0, 0);                    \
        }                                                                                                                     \
    }
#define Q_STAGE_P(H, U) { if (PB) Q_STAGE_B(H, U) else Q_STAGE_A(H, U) }
#define Q_STAGE_S(H, U) { if (PB) Q_STAGE_A(H, U) else Q_STAGE_B(H, U) }
#define Q_READ_A(H, T)                                                                                                        \
    {                                                                                                                         \
        const char* p_ = a_rd + ((T) & 1) * BUF + (H) * A_HALF;                                                               \
        _Pragma("unroll") for (int i = 0; i < RI; ++i) { fa[0][i][0] = *(const q_half8*)(p_ + i * 2048 + foff0); fa[0][i][1] = *(const q_half8*)(p_ + i * 2048 + foff1); } \
    }
#define Q_READ_B(G, T)                                                                                                        \
    {                                                                                                                         \
        const char* p_ = b_rd + ((T) & 1) * BUF + (G) * B_HALF;                                                               \
        _Pragma("unroll") for (int j = 0; j < CJ; ++j) { fb[PB ? (G) : 0][j][0] = *(const q_half8*)(p_ + j * 2048 + foff0); fb[PB ? (G) : 0][j][1] = *(const q_half8*)(p_ + j * 2048 + foff1); } \
    }
#define Q_READ_P(H, T) { if (PB) Q_READ_B(H, T) else Q_READ_A(H, T) }
#define Q_READ_S(H, T) { if (PB) Q_READ_A(H, T) else Q_READ_B(H, T) }
    // quadrant (HP of the persistent operand, HS of the streamed one)
#define Q_MFMA(HP, HS)                                                                                                        \
    {                                                                                                                         \
        constexpr int h_ = PB ? (HS) : (HP), g_ = PB ? (HP) : (HS);                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                                        \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                      \
            _Pragma("unroll") for (int i = 0; i < RI; ++i)                                                                    \
                _Pragma("unroll") for (int j = 0; j < CJ; ++j)                                                                \
                    acc[h_][g_][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[PB ? g_ : 0][j][kk], fa[0][i][kk], acc[h_][g_][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    }
#define Q_BAR() __builtin_amdgcn_s_barrier()
#define Q_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    // RR = false (registers allow both halves of P):                      RR = true (256 x 320: 160 accumulators + 40 B registers; P0 is read twice):
    //   ph1 read P0*, S0   (P0,S0)  stage S1(t+1)                           ph1 read S0*, P0  (P0,S0)  stage P0(t+1)     * = issued first, retired by a
    //   ph2 read P1        (P1,S0)  stage P0(t+2)                           ph2 read P1+      (P1,S0)  stage S0(t+2)         counted lgkmcnt before the barrier
    //   ph3 read S1        (P1,S1)  stage S0(t+2)                           ph3 read S1+      (P1,S1)  stage P1(t+2)     + = lgkmcnt(0) before the barrier: the
    //   ph4 --             (P0,S1)  stage P1(t+2), vmcnt(NW)                ph4 read P0+      (P0,S1)  stage S1(t+2), vmcnt(NW)   half-tile is restaged next phase
#define Q_TILE(T)                                                                                                             \
    {                                                                                                                         \
        /* phase 1 */                                                                                                         \
        if (RR) { Q_READ_S(0, T); __builtin_amdgcn_sched_barrier(0); Q_READ_P(0, T); }                                        \
        else { Q_READ_P(0, T); __builtin_amdgcn_sched_barrier(0); Q_READ_S(0, T); }                                           \
        if ((T) + 1 < nt) { if (RR) Q_STAGE_P(0, (T) + 1) else Q_STAGE_S(1, (T) + 1) }   /* scalars of tile T+1: set in phase 2 of tile T-1 / the prologue */ \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RR ? NRP : NRS) : "memory");                                               \
        Q_BAR();                                                                                                              \
        Q_MFMA(0, 0);                                                                                                         \
        Q_BAR();                                                                                                              \
        /* phase 2 */                                                                                                         \
        Q_READ_P(1, T);                                                                                                       \
        if ((T) + 2 < nt) { Q_SCAL((T) + 2); if (RR) Q_STAGE_S(0, (T) + 2) else Q_STAGE_P(0, (T) + 2) }                       \
        if (RR) Q_LGKM0();                                                                                                    \
        Q_BAR();                                                                                                              \
        Q_MFMA(1, 0);                                                                                                         \
        Q_BAR();                                                                                                              \
        /* phase 3 */                                                                                                         \
        Q_READ_S(1, T);                                                                                                       \
        if ((T) + 2 < nt) { if (RR) Q_STAGE_P(1, (T) + 2) else Q_STAGE_S(0, (T) + 2) }                                        \
        if (RR) Q_LGKM0();                                                                                                    \
        Q_BAR();                                                                                                              \
        Q_MFMA(1, 1);                                                                                                         \
        Q_BAR();                                                                                                              \
        /* phase 4 */                                                                                                         \
        if (RR) Q_READ_P(0, T);                                                                                               \
        if ((T) + 2 < nt) { if (RR) Q_STAGE_S(1, (T) + 2) else Q_STAGE_P(1, (T) + 2); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory"); } \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
        if (RR) Q_LGKM0();                                                                                                    \
        Q_BAR();                                                                                                              \
        Q_MFMA(0, 1);                                                                                                         \
        Q_BAR();                                                                                                              \
    }

    // prologue: tile 0 whole, then the first three half-tiles of tile 1 in the steady-state order; the fourth goes out in phase 1 of tile 0 with
    // tile 1's scalars still set
    Q_SCAL(0);
    if (RR) { Q_STAGE_S(0, 0); Q_STAGE_P(1, 0); Q_STAGE_S(1, 0); Q_STAGE_P(0, 0); }
    else { Q_STAGE_P(0, 0); Q_STAGE_S(0, 0); Q_STAGE_P(1, 0); Q_STAGE_S(1, 0); }
    if (nt > 1) {
        Q_SCAL(1);
        if (RR) { Q_STAGE_S(0, 1); Q_STAGE_P(1, 1); Q_STAGE_S(1, 1); }
        else { Q_STAGE_P(0, 1); Q_STAGE_S(0, 1); Q_STAGE_P(1, 1); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Q_BAR();
    if (grp == 1) Q_BAR();                        // group 1 runs one barrier behind
    int t = 0;
    if (!(G8Q_ABL & 4)) {
    for (; t + 1 < nt; t += 2) { Q_TILE(t); Q_TILE(t + 1); }       // t even: buffer parities are compile-time constants
    if (t < nt) Q_TILE(t);
    }
    if (grp == 0) Q_BAR();
#undef Q_TILE
#undef Q_LGKM0
#undef Q_MFMA
#undef Q_READ_S
#undef Q_READ_P
#undef Q_READ_B
#undef Q_READ_A
#undef Q_STAGE_S
#undef Q_STAGE_P
#undef Q_STAGE_B
#undef Q_STAGE_A
#undef Q_SCAL

    // ---- epilogue, straight from the accumulators (no LDS).  A lane holds, per 16 x 16 block (i, j), row (lane & 15) and the 4 consecutive columns
    // 4 k .. 4 k + 3 (k = lane >> 4): packed to two dwords.  v_permlane16_swap on the dwords of the blocks (2q, j) and (2q + 1, j) trades rows of 16
    // lanes -- X.row1 <-> Y.row0, X.row3 <-> Y.row2 -- after which lane group k' holds EIGHT consecutive columns (8 (k' >> 1) .. + 8) of row
    // (2q + (k' & 1)) * 16 + (lane & 15): one 16-B store (+ one 16-B residual load) per block pair.  Round 4's first version transposed every quadrant
    // through a wave-private LDS region (40 ds_write_b64 + 20 ds_read_b128 + their address arithmetic per lane): ~11 us per 256 x 320 tile, as much as
    // the five K tiles of a K = 320 Linear (tools/micro/gemm8_lab -DG8Q_ABL=1 on 368 640 x 320 x 64).  Same rounding points as every other configuration.
    if (G8Q_ABL & 2) { if (acc[0][0][0][0][0] == 12345.f) C[0] = (_Float16)1.f; return; }
    static_assert(RI % 2 == 0, "block pairs");
    const int mw0 = m0 + wr * 2 * QM, nw0 = n0 + wc * 2 * QN, l15 = lane & 15, kq = lane >> 4, l4 = kq * 4;
    const int sel = kq & 1, c8 = (kq >> 1) * 8;                  // after the swap: which block of the pair, which 8-column half
    typedef unsigned q_u32x4 __attribute__((ext_vector_type(4)));
    auto pk2 = [](float a, float b2) { const __attribute__((ext_vector_type(2))) _Float16 hh = {(_Float16)a, (_Float16)b2}; return __builtin_bit_cast(unsigned, hh); };
    auto swap_store = [&](unsigned x0, unsigned x1, unsigned y0, unsigned y1, int m, int n, int ldo, bool with_resid) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false), s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        q_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        if (m < M) {
            q_half8 v = __builtin_bit_cast(q_half8, o);
            if (with_resid) {
                const q_half8 rv = *(const q_half8*)(resid + (long)m * ldr + n);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (_Float16)post_act((float)v[q] + (float)rv[q], act);
            }
            if (!(G8Q_ABL & 1) || v[0] == (_Float16)12345.f) *(q_half8*)(C + (long)m * ldo + n) = v;
        }
    };
    if (act == 2) {                               // GEGLU (QN == 32): quadrant g = 0 holds the 32 values, g = 1 the matching gates of one 64-column group
        if constexpr (QN == 32) {
            q_half4 bvv[CJ], bvg[CJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                bvv[j] = bias ? *(const q_half4*)(bias + nw0 + j * 16 + l4) : q_half4{0, 0, 0, 0};
                bvg[j] = bias ? *(const q_half4*)(bias + nw0 + 32 + j * 16 + l4) : q_half4{0, 0, 0, 0};
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < RI / 2; ++q)
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            float o[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const _Float16 va = (_Float16)(acc[h][0][2 * q + e][j][r] + (float)bvv[j][r]), vg = (_Float16)(acc[h][1][2 * q + e][j][r] + (float)bvg[j][r]);
                                o[r] = (float)va * gelu_erf((float)vg);
                            }
                            pk[e][0] = pk2(o[0], o[1]); pk[e][1] = pk2(o[2], o[3]);
                        }
                        swap_store(pk[0][0], pk[0][1], pk[1][0], pk[1][1], mw0 + h * QM + (2 * q + sel) * 16 + l15, (nw0 >> 1) + j * 16 + c8, ldc, false);
                    }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        q_half4 bv[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) bv[j] = bias ? *(const q_half4*)(bias + nw0 + g * QN + j * 16 + l4) : q_half4{0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < RI / 2; ++q)
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const q_float4 a4 = acc[h][g][2 * q + e][j];
                        pk[e][0] = pk2(apply_act(a4[0] + (float)bv[j][0], act), apply_act(a4[1] + (float)bv[j][1], act));
                        pk[e][1] = pk2(apply_act(a4[2] + (float)bv[j][2], act), apply_act(a4[3] + (float)bv[j][3], act));
                    }
                    swap_store(pk[0][0], pk[0][1], pk[1][0], pk[1][1], mw0 + h * QM + (2 * q + sel) * 16 + l15, nw0 + g * QN + j * 16 + c8, ldc, resid != nullptr);
                }
    }
#endif
}

template <int WM, int WN, int RI, int CJ>
static int launch8q(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K, int lda,
                    int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st) {
    constexpr int QN = CJ * 16, BN = 2 * WN * QN, B_HALF = WN * QN * 128, BM = 2 * WM * RI * 16, A_HALF = WM * RI * 16 * 128;
    const int tm = cdiv(M, BM), tn = N / BN;
    const size_t lds = (size_t)2 * (2 * A_HALF + 2 * B_HALF) + (B_HALF % 8192 ? 4096 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_gemm8q<WM, WN, RI, CJ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8q<WM, WN, RI, CJ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8q<WM, WN, RI, CJ, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const size_t a_bytes = cp.conv ? (size_t)(M / (cp.Hout * cp.Wout)) * cp.Hin * cp.Win * cp.Cin * 2 : ((size_t)(M - 1) * lda + K) * 2;
    const size_t w_bytes = ((size_t)(N - 1) * ldw + K) * 2;
    if (cp.conv && (cp.Hup != cp.Hin || cp.Wup != cp.Win)) hipLaunchKernelGGL((k_gemm8q<WM, WN, RI, CJ, true, true>), dim3(tm * tn), dim3(512), lds, st, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, tm, tn, (unsigned long)a_bytes, (unsigned)w_bytes);
    else if (cp.conv) hipLaunchKernelGGL((k_gemm8q<WM, WN, RI, CJ, true>), dim3(tm * tn), dim3(512), lds, st, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, tm, tn, (unsigned long)a_bytes, (unsigned)w_bytes);
    else hipLaunchKernelGGL((k_gemm8q<WM, WN, RI, CJ, false>), dim3(tm * tn), dim3(512), lds, st, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, tm, tn, (unsigned long)a_bytes, (unsigned)w_bytes);
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

// Can the 8-phase kernel take this call?  cfg 1 = 256 x 256 (N % 256 == 0; GEGLU allowed), 2 = 256 x 320 (N % 320 == 0), 3 = 512 x 128 (N % 128 == 0:
// the VAE's 128-channel convolutions at full resolution; WM 8, WN 1, quadrant 32 x 64, all 160 KiB of LDS).  K % 64 == 0 (conv:
// Cin % 64 == 0), the weights and one tile's rows of A addressable with 32 bits (A as a whole may be larger), nearest up-sampling only by a factor <= 2 per axis (stride 1, pad 1), 16-B aligned rows.
bool gemm8q_ok(int cfg, int M, int N, int K, int lda, int ldw, int ldc, int ldr, bool has_resid, int act, const ConvP& cp) {
    const int BN = cfg == 1 ? 256 : (cfg == 2 ? 320 : 128);
    if (N % BN || K % 64 || K < 64 || M < 1) return false;
    if ((ldw & 7) || (ldc & 7) || (has_resid && (ldr & 7))) return false;
    if (act == 2 && (cfg != 1 || has_resid)) return false;
    if (act < 0 || act > 5) return false;
    if (cp.conv) {
        if (cp.Cin % 64) return false;
        if (cp.Hup != cp.Hin || cp.Wup != cp.Win) {      // nearest up-sampling in the gather: stride 1, pad 1, scale in (0.5, 1] per axis (source-row deltas 0..2)
            if (cp.stride != 1 || cp.pad != 1 || cp.Hup < cp.Hin || cp.Wup < cp.Win || cp.Hup > 2 * cp.Hin || cp.Wup > 2 * cp.Win) return false;
        }
        if ((size_t)4 * cp.Hin * cp.Win * cp.Cin * 2 >= 0xffffff00ull) return false;      // a tile's rows touch <= 3 consecutive images (addressed relative to the first)
    } else {
        if ((lda & 7) || (size_t)256 * 2 * lda * 2 >= 0xffffff00ull) return false;
    }
    return ((size_t)(N - 1) * ldw + K) * 2 < 0xffffff00ull;
}

int gemm8q_dispatch(int cfg, const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                    int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st) {
    if (!gemm8q_ok(cfg, M, N, K, lda, ldw, ldc, ldr, resid != nullptr, act, cp)) return TCL_EINVAL;
    if (cfg == 1) return launch8q<2, 4, 4, 2>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    if (cfg == 3) return launch8q<8, 1, 2, 4>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    return launch8q<4, 2, 2, 5>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
}
