// 8-wave "ping-pong" fp16 MFMA GEMM / implicit 3x3 convolution for gfx950: the large-problem path behind tcl_gemm_f16 /
// tcl_conv3x3_f16 (same contract as gemm.hip: C = act(A.W^T + bias) + resid, f32 accumulate; reference call sites
// generate.py:342-347 -> diffusers UNet2DConditionModel convs / Linears).
//
// Block = 512 threads = 8 waves = exactly two waves per SIMD, one block per CU.  Block tile (WM*MT*32) x (WN*NT*32), K step 32.
// The waves form two groups (waves 0-3 / 4-7: wave w and w+4 share a SIMD) that run the SAME program shifted by one barrier
// interval, so in every interval one group issues its MFMAs while the other one does its LDS fragment reads and issues the
// LDS-DMA (global_load_lds_dwordx4) pieces of a later stage: the matrix pipe of every SIMD is fed back to back and the
// loads/ds_reads ride in its shadow.  Per wave and K step t:
//      G0:        { bar  L(t)         bar  M(t) W(t+1) } x nk   bar
//      G1:  bar   { bar  L(t) W(t+1)  bar  M(t)        } x nk          (2 nk + 1 barriers each)
//   L(t): ds_read the 2 x (MT+NT) b128 fragments of stage t, issue the NP DMA pieces of stage t+PD, s_waitcnt lgkmcnt(0)
//   M(t): 2 x MT x NT MFMA 32x32x16 on the fragments (s_setprio 1)
//   W(t+1): s_waitcnt vmcnt((PD-1)*NP)  -- my pieces of stage t+1 have landed; later stages stay in flight (never drained)
// Invariants (b_k = k-th block barrier; G0 runs L(t) in (b_2t, b_2t+1), G1 in (b_2t+1, b_2t+2)):
//   RAW  every wave executes W(t) before b_2t, every read of stage t comes after b_2t.
//   WAR  stage t+PD reuses the ring slot of stage t-1 (PD+1 slots); the last reads of stage t-1 (G1, L(t-1)) are retired by
//        the lgkmcnt(0) that precedes b_2t; the earliest DMA into that slot is issued by G0 in L(t), after b_2t.
// LDS rows are 64 B (32 halves), 16-B chunk index XOR-swizzled with (row>>2)&3 on the DMA source address and on the ds_read
// address (conflict-free b128 reads, guide rule 21).  Out-of-range rows / conv taps read a zero page.
// Epilogue: each wave stages its own 32-row strips through a private LDS region (no block barrier) and writes whole 16-B
// row chunks (+ residual).
#include "common.h"
#include "../../include/tclight_hip.h"
#include "gemm_conv.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) unsigned g_zero_page8[64];
#ifdef G8_PROF
__device__ unsigned long long* g8_prof_buf;
#endif
#ifndef G8_ABL
#define G8_ABL 0      // lab-only knock-outs of k_gemm8p: 1 = no DMA in the K loop, 2 = no fragment reads in the K loop (results are garbage)
#endif

template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// Epilogue shared by both schedules: each wave stages its own 32-row strips through a private LDS region (no block barrier; all ring
// reads retired and no DMA in flight) and writes whole 16-B row chunks (+ residual / GEGLU).
template <int MT, int NT, int WM, int WN>
__device__ __forceinline__ void g8_epilogue(float16v (&acc)[MT][NT], char* smem, const _Float16* __restrict__ bias, const _Float16* __restrict__ resid,
                                            _Float16* __restrict__ C, int M, int N, int ldc, int ldr, int act, int m0, int n0, int wid, int lane, int wm, int wn) {
    constexpr int WCOLS = NT * 32, CSW = WCOLS + 8, CPRW = WCOLS / 8;
    _Float16* Cs = (_Float16*)smem + wid * 32 * CSW;
    float bv[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) { const int n = n0 + wn * WCOLS + b * 32 + (lane & 31); bv[b] = (bias && n < N) ? (float)bias[n] : 0.f; }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][b][r] + bv[b];
                v = apply_act(v, act);
                Cs[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CSW + b * 32 + (lane & 31)] = (_Float16)v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (act == 2) {            // GEGLU (NT == 2 only): the wave's 64 columns are one [32 value | 32 gate] group -> 32 output columns
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = lane + 64 * i, row = c >> 2, c8 = (c & 3) * 8, m = m0 + (wm * MT + a) * 32 + row, n = ((n0 + wn * WCOLS) >> 1) + c8;
                if (m < M && n < (N >> 1)) {
                    half8 va = *(const half8*)(Cs + row * CSW + c8), vg = *(const half8*)(Cs + row * CSW + 32 + c8);
#pragma unroll
                    for (int q = 0; q < 8; ++q) { float gf = (float)vg[q]; va[q] = (_Float16)((float)va[q] * gelu_erf(gf)); }
                    *(half8*)(C + (long)m * ldc + n) = va;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 32 * CPRW / 64; ++i) {
            if (act == 2) break;
            const int c = lane + 64 * i, row = c / CPRW, c8 = (c % CPRW) * 8, m = m0 + (wm * MT + a) * 32 + row, n = n0 + wn * WCOLS + c8;
            if (m < M && n < N) {
                half8 v = *(const half8*)(Cs + row * CSW + c8);
                if (resid) {
                    half8 rv = *(const half8*)(resid + (long)m * ldr + n);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (_Float16)post_act((float)v[q] + (float)rv[q], act);
                }
                *(half8*)(C + (long)m * ldc + n) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MT, int NT, int WM, int WN, bool CONV>
__global__ __launch_bounds__(512) void k_gemm8(const _Float16* __restrict__ A, const _Float16* __restrict__ W, const _Float16* __restrict__ bias,
                                               const _Float16* __restrict__ resid, _Float16* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                               int ldc, int ldr, int act, ConvP cp, int tiles_m, int tiles_n) {
    static_assert(WM * WN == 8, "8 waves");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, KB = 32, ROWB = 64;
    constexpr int NS = 4;                                      // ring slots: stage t in use, t+1 .. t+3 in flight
    constexpr int A_P = BM / 16, B_P = BN / 16;                // 1-KiB pieces (16 rows x 64 B) per operand and stage
    constexpr int NP = (A_P + B_P + 7) / 8;                    // pieces per wave and stage (a dummy piece pads the last round)
    constexpr int NPA = A_P / 8, NPB = NP - NPA;               // rounds that carry A pieces (A_P % 8 == 0) / W pieces
    constexpr int STAGE = (A_P + B_P) * 1024;
    static_assert(A_P % 8 == 0, "BM must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];     // NS stages, then a 1-KiB dump for the dummy pieces

    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int tn = j % tiles_n, tm = (j / tiles_n) * 8 + xcd;
    if (tm >= tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wid >> 2;
    const int wm = wid % WM, wn = wid / WM;

    // ---- DMA descriptors: round i handles piece q = wid + 8 i; lane -> row rr = lane>>2 of the piece, LDS chunk lane&3.
    // Invalid rows (m >= M, n >= N) and the dummy piece point at the zero page.
    const int rr = lane >> 2, csrc = ((lane & 3) ^ ((rr >> 2) & 3)) * 8;
    const _Float16* zero = (const _Float16*)g_zero_page8;
    const _Float16* ap[NPA]; int a_oy[NPA], a_ox[NPA];          // dense: row pointer (k advances); conv: image base + (oy, ox)
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + (wid + 8 * i) * 16 + rr;
        if (!CONV) { ap[i] = m < M ? A + (long)m * lda + csrc : nullptr; a_oy[i] = a_ox[i] = 0; }
        else {
            int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
            ap[i] = A + (long)b * cp.Hin * cp.Win * cp.Cin + csrc;
            a_oy[i] = m < M ? oy * cp.stride - cp.pad : -(1 << 20); a_ox[i] = ox * cp.stride - cp.pad;
        }
    }
    const _Float16* wp[NPB]; int wdst[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int q = wid + 8 * (i + NPA), n = n0 + (q - A_P) * 16 + rr;
        const bool real = q < A_P + B_P;
        wp[i] = (real && n < N) ? W + (long)n * ldw + csrc : nullptr;
        wdst[i] = real ? q * 1024 : NS * STAGE;                  // dummy piece -> dump area
    }

    // issue the pieces of stage KT, then those of stage KT+1 (when < nk): the second 64-B halves of the same 128-B lines follow
    // within a few instructions, while the first requests are still pending in the vector L1
#define G8_GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
#define G8_ISSUE2(KT)                                                                                                         \
    {                                                                                                                         \
        int k0_ = (KT) * KB;                                                                                                  \
        const int nst_ = (KT) + 1 < nk ? 2 : 1;                                                                               \
        int tdy_ = 0, tdx_ = 0, c0_ = k0_;                                                                                    \
        if (CONV) { const int tap_ = conv_kmap(k0_, cp.Cin, c0_); tdy_ = tap_ / 3; tdx_ = tap_ - tdy_ * 3; k0_ = tap_ * cp.Cin + c0_; }   \
        const _Float16* sa_[NPA];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                                    \
            if (!CONV) sa_[i] = ap[i] ? ap[i] + k0_ : nullptr;                                                                \
            else {                                                                                                            \
                int iy_ = a_oy[i] + tdy_, ix_ = a_ox[i] + tdx_;                                                               \
                const bool in_ = iy_ >= 0 && iy_ < cp.Hup && ix_ >= 0 && ix_ < cp.Wup;                                        \
                if (cp.Hup != cp.Hin || cp.Wup != cp.Win) { iy_ = min((int)floorf(iy_ * cp.sy), cp.Hin - 1); ix_ = min((int)floorf(ix_ * cp.sx), cp.Win - 1); } \
                sa_[i] = in_ ? ap[i] + ((long)iy_ * cp.Win + ix_) * cp.Cin + c0_ : nullptr;                                   \
            }                                                                                                                 \
        }                                                                                                                     \
        for (int h_ = 0; h_ < nst_; ++h_) {                                                                                   \
            char* sb_ = smem + (((KT) + h_) % NS) * STAGE;                                                                    \
            _Pragma("unroll") for (int i = 0; i < NPA; ++i) G8_GLDS(sa_[i] ? sa_[i] + h_ * KB : zero, sb_ + (wid + 8 * i) * 1024); \
            _Pragma("unroll") for (int i = 0; i < NPB; ++i) G8_GLDS(wp[i] ? wp[i] + k0_ + h_ * KB : zero, (wdst[i] == NS * STAGE ? smem : sb_) + wdst[i]); \
        }                                                                                                                     \
    }

    float16v acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / KB;
    const int frow = lane & 31, fh = lane >> 5;
    half8 fa[2][MT], fb[2][NT];
    // fragment read: tile row R (lane&31 within a 32-row block), logical chunk 2*ks + (lane>>5), physical chunk ^ ((R>>2)&3)
#define G8_READ(KT)                                                                                                           \
    {                                                                                                                         \
        const char* ab_ = smem + ((KT) % NS) * STAGE + (wm * MT * 32) * ROWB;                                                 \
        const char* bb_ = smem + ((KT) % NS) * STAGE + BM * ROWB + (wn * NT * 32) * ROWB;                                     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                   \
            _Pragma("unroll") for (int a = 0; a < MT; ++a) { int R = a * 32 + frow; fa[ks][a] = *(const half8*)(ab_ + R * ROWB + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); } \
            _Pragma("unroll") for (int b = 0; b < NT; ++b) { int R = b * 32 + frow; fb[ks][b] = *(const half8*)(bb_ + R * ROWB + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); } \
        }                                                                                                                     \
    }
#define G8_MFMA()                                                                                                             \
    {                                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                      \
            _Pragma("unroll") for (int a = 0; a < MT; ++a)                                                                    \
                _Pragma("unroll") for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][a], fb[ks][b], acc[a][b], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                        \
    }
    // W(T) executed at the end of step T-1's interval: my pieces of stage T have landed.  Issued so far: stages <= T+2 when T-1 is
    // even (the pair (T+1, T+2) went out in L(T-1)), <= T+1 when T-1 is odd; later stages stay in flight.
#define G8_WAIT(T)                                                                                                            \
    {                                                                                                                         \
        const int rem_ = min(((T) & 1) ? (T) + 2 : (T) + 1, nk - 1) - (T);                                                    \
        if (rem_ >= 2) wait_vm<2 * NP>(); else if (rem_ == 1) wait_vm<NP>(); else wait_vm<0>();                               \
    }
#define G8_LSEG(T)                                                                                                            \
    {                                                                                                                         \
        G8_READ(T);                                                                                                           \
        if (!((T) & 1) && (T) + 2 < nk) G8_ISSUE2((T) + 2);                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    }

    G8_ISSUE2(0);
    if (nk > 1) wait_vm<NP>(); else wait_vm<0>();              // W(0): stage 1 may still be in flight
    if (grp == 0) {
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            G8_LSEG(t);
            __builtin_amdgcn_s_barrier();
            G8_MFMA();
            if (t + 1 < nk) G8_WAIT(t + 1);
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();                 // one interval behind group 0
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            G8_LSEG(t);
            if (t + 1 < nk) G8_WAIT(t + 1);
            __builtin_amdgcn_s_barrier();
            G8_MFMA();
        }
    }
#undef G8_ISSUE2
#undef G8_GLDS
#undef G8_READ
#undef G8_MFMA
#undef G8_WAIT
#undef G8_LSEG

    g8_epilogue<MT, NT, WM, WN>(acc, smem, bias, resid, C, M, N, ldc, ldr, act, m0, n0, wid, lane, wm, wn);
}

// ---------------------------------------------------------------------------------------------------------------------
// Schedule 2 ("DMA in the MFMA shadow", round 3).  Same tiles, ring, swizzle, K order and epilogue as k_gemm8 -- so the results are
// bit-identical -- but the LDS-DMA pieces of the stage PD = 3 steps ahead are issued INSIDE the MFMA segment, one piece every
// nM / PM matrix instructions, with their source addresses computed beforehand in the (short) LDS-read segment:
//      G0:        { bar  L(t) P(t+3)          bar  M(t)+D(t+3) W0(t+1) } x nk   bar
//      G1:  bar   { bar  L(t) P(t+3) W1(t+1)  bar  M(t)+D(t+3)         } x nk
//   P(s): per-lane source pointers of my NP pieces of stage s (VALU / SALU only; conv: tap, bounds, nearest-upsample mapping)
//   D(s): the NP glds instructions (the first NL of them already in the L segment), pinned between the MFMAs (sched_barrier)
// Measured reason (r2 PMC: waves 52 % parked, matrix pipe 27-50 %): an LDS-DMA piece costs 100-185 issue cycles inside a segment that
// also carries 12-14 ds_read_b128, so the L segment (5 pieces + reads) was longer than the partner's 16-20 MFMAs and the matrix pipe
// waited for it; among bare MFMAs a piece costs ~60 cycles, half of which the running MFMA covers.
// Invariants (G0 runs L(t) in I(2t), M(t) in I(2t+1); G1 one interval later; I(k) = (b_k, b_k+1)):
//   WAR  stage t+3 takes the ring slot of stage t-1, last read by G1 in L(t-1) = I(2t-1) and retired (lgkmcnt 0) before b_2t; the
//        earliest piece of stage t+3 is issued in G0's L(t) = I(2t).
//   RAW  every wave passes W(t+1) ("my pieces of stage t+1 have landed") before b_2t+2; the first read of stage t+1 is G0's L(t+1)
//        = I(2t+2).  W0 runs after D(t+3): up to two whole stages stay in flight; W1 runs before it: stage t+2 and the NL early pieces.
template <int MT, int NT, int WM, int WN, bool CONV>
__global__ __launch_bounds__(512) void k_gemm8s(const _Float16* __restrict__ A, const _Float16* __restrict__ W, const _Float16* __restrict__ bias,
                                                const _Float16* __restrict__ resid, _Float16* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                int ldc, int ldr, int act, ConvP cp, int tiles_m, int tiles_n) {
    static_assert(WM * WN == 8, "8 waves");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, KB = 32, ROWB = 64;
    constexpr int NS = 4, PD = 3;
    constexpr int A_P = BM / 16, B_P = BN / 16;
    constexpr int NP = (A_P + B_P + 7) / 8;
    constexpr int NPA = A_P / 8, NPB = NP - NPA;
    constexpr int STAGE = (A_P + B_P) * 1024;
    constexpr int NL = 0;                                  // pieces of a stage still issued in the read segment (measured: 0 is best)
    constexpr int nM = 2 * MT * NT, PM = NP - NL, G = nM / (PM > 0 ? PM : 1);
    static_assert(A_P % 8 == 0 && G >= 1, "tile / schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int tn = j % tiles_n, tm = (j / tiles_n) * 8 + xcd;
    if (tm >= tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wid >> 2;
    const int wm = wid % WM, wn = wid / WM;

    const int rr = lane >> 2, csrc = ((lane & 3) ^ ((rr >> 2) & 3)) * 8;
    const _Float16* zero = (const _Float16*)g_zero_page8;
    const _Float16* ap[NPA]; int a_oy[NPA], a_ox[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + (wid + 8 * i) * 16 + rr;
        if (!CONV) { ap[i] = m < M ? A + (long)m * lda + csrc : nullptr; a_oy[i] = a_ox[i] = 0; }
        else {
            int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
            ap[i] = A + (long)b * cp.Hin * cp.Win * cp.Cin + csrc;
            a_oy[i] = m < M ? oy * cp.stride - cp.pad : -(1 << 20); a_ox[i] = ox * cp.stride - cp.pad;
        }
    }
    const _Float16* wp[NPB]; int wdst[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int q = wid + 8 * (i + NPA), n = n0 + (q - A_P) * 16 + rr;
        const bool real = q < A_P + B_P;
        wp[i] = (real && n < N) ? W + (long)n * ldw + csrc : nullptr;
        wdst[i] = real ? q * 1024 : -1;
    }
    const int nk = K / KB;
    const _Float16* src[NP];
#define G8S_GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
    // P(KT): source pointers of my pieces of stage KT
#define G8S_PREP(KT)                                                                                                          \
    {                                                                                                                         \
        int k0_ = (KT) * KB;                                                                                                  \
        const bool ex_ = (KT) < nk;       /* stages past the end: every piece reads the zero page into the dump area */         \
        int tdy_ = 0, tdx_ = 0, c0_ = k0_;                                                                                    \
        if (CONV) { const int tap_ = conv_kmap(k0_, cp.Cin, c0_); tdy_ = tap_ / 3; tdx_ = tap_ - tdy_ * 3; k0_ = tap_ * cp.Cin + c0_; }   \
        _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                                    \
            if (!CONV) src[i] = (ex_ && ap[i]) ? ap[i] + k0_ : zero;                                                                   \
            else {                                                                                                            \
                int iy_ = a_oy[i] + tdy_, ix_ = a_ox[i] + tdx_;                                                               \
                const bool in_ = iy_ >= 0 && iy_ < cp.Hup && ix_ >= 0 && ix_ < cp.Wup;                                        \
                if (cp.Hup != cp.Hin || cp.Wup != cp.Win) { iy_ = min((int)floorf(iy_ * cp.sy), cp.Hin - 1); ix_ = min((int)floorf(ix_ * cp.sx), cp.Win - 1); } \
                src[i] = (ex_ && in_) ? ap[i] + ((long)iy_ * cp.Win + ix_) * cp.Cin + c0_ : zero;                                   \
            }                                                                                                                 \
        }                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < NPB; ++i) src[NPA + i] = (ex_ && wp[i]) ? wp[i] + k0_ : zero;                         \
    }
    // D(KT, i): piece i of stage KT (dummy pieces of the last round land in the 1-KiB dump area behind the ring)
#define G8S_FIRE(KT, I)                                                                                                       \
    {                                                                                                                         \
        char* sb_ = smem + ((KT) % NS) * STAGE;                                                                               \
        char* dump_ = smem + NS * STAGE;                                                                                      \
        if ((I) < NPA) G8S_GLDS(src[(I)], (KT) < nk ? sb_ + (wid + 8 * (I)) * 1024 : dump_);                                  \
        else G8S_GLDS(src[(I)], ((KT) < nk && wdst[(I) - NPA] >= 0) ? sb_ + wdst[(I) - NPA] : dump_);                         \
    }

    float16v acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    half8 fa[2][MT], fb[2][NT];
#define G8S_READ(KT)                                                                                                          \
    {                                                                                                                         \
        const char* ab_ = smem + ((KT) % NS) * STAGE + (wm * MT * 32) * ROWB;                                                 \
        const char* bb_ = smem + ((KT) % NS) * STAGE + BM * ROWB + (wn * NT * 32) * ROWB;                                     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                   \
            _Pragma("unroll") for (int a = 0; a < MT; ++a) { int R = a * 32 + frow; fa[ks][a] = *(const half8*)(ab_ + R * ROWB + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); } \
            _Pragma("unroll") for (int b = 0; b < NT; ++b) { int R = b * 32 + frow; fb[ks][b] = *(const half8*)(bb_ + R * ROWB + (((2 * ks + fh) ^ ((R >> 2) & 3)) << 4)); } \
        }                                                                                                                     \
    }
    // fragments of K substep KS (0 | 1) of stage KT only
#define G8S_READK(KT, KS)                                                                                                     \
    {                                                                                                                         \
        const char* ab_ = smem + ((KT) % NS) * STAGE + (wm * MT * 32) * ROWB;                                                 \
        const char* bb_ = smem + ((KT) % NS) * STAGE + BM * ROWB + (wn * NT * 32) * ROWB;                                     \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) { int R = a * 32 + frow; fa[KS][a] = *(const half8*)(ab_ + R * ROWB + (((2 * (KS) + fh) ^ ((R >> 2) & 3)) << 4)); } \
        _Pragma("unroll") for (int b = 0; b < NT; ++b) { int R = b * 32 + frow; fb[KS][b] = *(const half8*)(bb_ + R * ROWB + (((2 * (KS) + fh) ^ ((R >> 2) & 3)) << 4)); } \
    }
    // M(t) with the PM late pieces of stage KT pinned between the MFMAs: piece NL + q right after MFMA q * G
#define G8S_MSEG(KT) G8S_MRANGE(KT, 0, nM)
#define G8S_MRANGE(KT, LO, HI) G8S_MRANGEF(KT, LO, HI, 0)
#define G8S_MRANGEF(KT, LO, HI, FOFF)                                                                                                 \
    {                                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                                        \
        _Pragma("unroll") for (int mi = (LO); mi < (HI); ++mi) {                                                             \
            const int ks_ = mi / (MT * NT), a_ = (mi / NT) % MT, b_ = mi % NT;                                                \
            acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks_][a_], fb[ks_][b_], acc[a_][b_], 0, 0, 0);            \
            if (PM > 0 && ((mi + (FOFF)) % nM) % G == 0 && ((mi + (FOFF)) % nM) / G < PM) {                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                            \
                G8S_FIRE(KT, NL + ((mi + (FOFF)) % nM) / G);                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                                            \
            }                                                                                                                 \
        }                                                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                                        \
    }
    // W0(T): after D(T+2); W1(T): before it (only its NL early pieces are out)
    // W0 runs after D(t+3): two whole stages may stay in flight; W1 before it: stage t+2 and the NL early pieces of stage t+3.  The counts
    // are constants because stages past the end are issued as dummies.
#define G8S_WAIT0() wait_vm<2 * NP>()
#define G8S_WAIT1() wait_vm<NP + NL>()

    // prologue: stages 0 .. PD-1 whole (dummies when nk < PD)
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        G8S_PREP(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) G8S_FIRE(s, i);
    }
    wait_vm<2 * NP>();                                  // W(0)
#ifdef G8_PROF      // lab only (tools/micro/gemm8_lab.hip): shader cycles per segment, summed over the K loop, for wave 0 / wave 4 of each block
    unsigned long long tsum_[5] = {0, 0, 0, 0, 0}, tlast_, tnow_;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tlast_));
#define G8S_T(I) { asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tnow_) :: "memory"); tsum_[I] += tnow_ - tlast_; tlast_ = tnow_; }
#else
#define G8S_T(I)
#endif
    if (grp == 0) {
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            G8S_T(0)
            G8S_READ(t);
            G8S_PREP(t + PD);
#pragma unroll
            for (int i = 0; i < NL; ++i) G8S_FIRE(t + PD, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G8S_T(1)
            __builtin_amdgcn_s_barrier();
            G8S_T(2)
            G8S_MSEG(t + PD);
            G8S_T(3)
            G8S_WAIT0();
            G8S_T(4)
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            G8S_T(0)
            G8S_READ(t);
            G8S_PREP(t + PD);
#pragma unroll
            for (int i = 0; i < NL; ++i) G8S_FIRE(t + PD, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G8S_T(1)
            G8S_WAIT1();
            G8S_T(4)
            __builtin_amdgcn_s_barrier();
            G8S_T(2)
            G8S_MSEG(t + PD);
            G8S_T(3)
        }
    }
#ifdef G8_PROF
    if (g8_prof_buf && lane == 0 && (wid & 3) == 0 && blockIdx.x < 64) {
#pragma unroll
        for (int i = 0; i < 5; ++i) g8_prof_buf[(blockIdx.x * 2 + grp) * 5 + i] = tsum_[i];
    }
#endif
#undef G8S_T
    wait_vm<0>();                                       // dummy pieces of the last steps (dump area) before the epilogue reuses the LDS
#undef G8S_GLDS
#undef G8S_PREP
#undef G8S_FIRE
#undef G8S_READ
#undef G8S_READK
#undef G8S_MRANGE
#undef G8S_MRANGEF
#undef G8S_MSEG
#undef G8S_WAIT0
#undef G8S_WAIT1
    g8_epilogue<MT, NT, WM, WN>(acc, smem, bias, resid, C, M, N, ldc, ldr, act, m0, n0, wid, lane, wm, wn);
}

// ---------------------------------------------------------------------------------------------------------------------
// Schedule 5 (k_gemm8p, round 3): software-pipelined half steps, ONE block barrier per K step, buffer-addressed LDS-DMA.  Same tiles,
// ring, swizzle, K order and epilogue as k_gemm8 (bit-identical results).
//
// Why (segment timers of tools/micro/gemm8_lab.hip on k_gemm8s, 256 x 320 tile, cycles per K step of 32): the two-barrier ping-pong
// spends 2 x (640 MFMA + 5 x 50 DMA issue) in the two MFMA segments plus 110-170 idle cycles per barrier hand-over, and the LDS-read +
// address segment (350 dense, 900-1100 conv) of one group has to fit under the other group's MFMA segment: 2 200-2 300 cycles against
// 1 280 of matrix work.  Here a wave never stops issuing MFMAs for its reads:
//   half step h = 2t + ks: the MT NT MFMAs of K substep ks of stage t run while the fragments of half step h + 1 are read into the
//   OTHER register half (free since half step h - 1) and the DMA pieces of stage t + 3 go out between the MFMAs.
//   B_t = "every wave has waited for its pieces of stage t + 1 and retired its reads of stage t - 1".  Group 0 passes B_t before half
//   step 2t, group 1 before half step 2t - 1: the groups stay half a step apart, so the two waves of a SIMD interleave their MFMA
//   streams freely instead of alternating through barrier hand-overs.
//   RAW  G0 reads stage t (ks 1) and stage t+1 (ks 0) after B_t; G1 reads stage t (ks 0, ks 1) after B_t: all covered by "stage t + 1
//        landed at B_t".  W = vmcnt(NP) right before a wave's barrier: only the stage it has just issued stays in flight.
//   WAR  stage t+3 -> ring slot of stage t-1, issued after B_t by both groups (G0: half steps 2t, 2t+1; G1: 2t-1, 2t).  Reads of stage
//        t-1 end in half step 2t-2 for both groups and are retired (lgkmcnt 0) before the wave's next barrier, which is B_t or earlier.
// Addressing: a DMA piece is `buffer_load_dwordx4 ... offen lds` through one resource descriptor per operand; the per-lane part of the
// address is ONE 32-bit byte offset fixed for the whole K loop (row of the piece + swizzled 16-B chunk), the K step adds a scalar.
//   * padding needs no zero page: offset 0xffffffff is outside the descriptor's range and the hardware returns zeros.  Rows past M / N are
//     clamped to the last row instead (their outputs are never stored);
//   * implicit 3x3 convolution (stride 1 | 2, no up-sampling): per lane and piece the base offset of output pixel (oy, ox) and a 9-bit
//     validity mask of its taps, computed once; per stage the tap / channel-slice offset is a scalar: 3 vector instructions per A piece
//     and stage instead of ~25 (k_gemm8s: iy / ix / bounds / 64-bit address per piece), and no per-stage pointer table in registers.
// Requires operands addressable with 32 bits (host: bytes < 4 GiB, else the k_gemm8s path; up-sampling convolutions take k_gemm8s too).
template <int MT, int NT, int WM, int WN, bool CONV>
__global__ __launch_bounds__(512) void k_gemm8p(const _Float16* __restrict__ A, const _Float16* __restrict__ W, const _Float16* __restrict__ bias,
                                                const _Float16* __restrict__ resid, _Float16* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                int ldc, int ldr, int act, ConvP cp, int tiles_m, int tiles_n, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist in the device pass only; the host pass needs just the stub
    static_assert(WM * WN == 8, "8 waves");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, KB = 32, ROWB = 64;
    constexpr int NS = 4, PD = 3;
    constexpr int A_P = BM / 16, B_P = BN / 16;
    constexpr int NP = (A_P + B_P + 7) / 8;
    constexpr int NPA = A_P / 8, NPB = NP - NPA;
    constexpr int STAGE = (A_P + B_P) * 1024;
    constexpr int nM = 2 * MT * NT, HALF = MT * NT, G = nM / NP;
    static_assert(A_P % 8 == 0 && G >= 1, "tile / schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int tn = j % tiles_n, tm = (j / tiles_n) * 8 + xcd;
    if (tm >= tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wid >> 2;
    const int wm = wid % WM, wn = wid / WM;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);
    const int rr = lane >> 2, csrc = ((lane & 3) ^ ((rr >> 2) & 3)) * 16;         // source chunk (bytes) of this lane's LDS chunk
    unsigned aoff[NPA], amask[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + (wid + 8 * i) * 16 + rr;
        if (!CONV) { aoff[i] = (unsigned)min(m, M - 1) * (unsigned)lda * 2u + csrc; amask[i] = 0; }
        else {
            const int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
            const int iy0 = oy * cp.stride - cp.pad, ix0 = ox * cp.stride - cp.pad;
            aoff[i] = (unsigned)(((b * cp.Hin + iy0) * cp.Win + ix0) * cp.Cin) * 2u + csrc;      // wraps for border pixels; only used with a valid tap
            unsigned mk = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = iy0 + tap / 3, ix = ix0 + tap % 3;
                if (m < M && iy >= 0 && iy < cp.Hin && ix >= 0 && ix < cp.Win) mk |= 1u << tap;
            }
            amask[i] = mk;
        }
    }
    unsigned woff[NPB]; int wdst[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int q = wid + 8 * (i + NPA), n = n0 + (q - A_P) * 16 + rr;
        const bool real = q < A_P + B_P;
        woff[i] = (unsigned)min(real ? n : n0, N - 1) * (unsigned)ldw * 2u + csrc;
        wdst[i] = real ? q * 1024 : -1;
    }
    const int nk = K / KB;
    // Scalars of the stage being issued, computed ONCE per (half) step by G8P_STAGE and shared by its NP pieces: ring slot, K offsets of the two
    // operands (conv: tap + channel slice -> input offset and weight column), whether the stage exists at all (stages past the end are issued
    // as dummies that re-read the last stage into the dump area, so that every wave issues NP pieces per stage and the vmcnt counts are
    // constants).  The first version recomputed all of it inside every piece, pinned between two MFMAs: 6.7 scalar instructions per MFMA
    // (rocprofv3 SQ_INSTS_SALU, profiles/r3_gemm8p_counters_*.txt).
    char* st_base = smem; unsigned st_ka = 0, st_kw = 0; int st_tap = 0; bool st_ex = false;
#define G8P_STAGE(KT)                                                                                                         \
    {                                                                                                                         \
        const int kt_ = min((KT), nk - 1);                                                                                    \
        st_ex = (KT) < nk; st_base = smem + ((KT) % NS) * STAGE;                                                              \
        int ka_ = kt_ * KB, kw_ = ka_; st_tap = 0;                                                                            \
        if (CONV) { int c0_; st_tap = conv_kmap(ka_, cp.Cin, c0_); kw_ = st_tap * cp.Cin + c0_; ka_ = ((st_tap / 3) * cp.Win + st_tap % 3) * cp.Cin + c0_; } \
        st_ka = (unsigned)ka_ * 2u; st_kw = (unsigned)kw_ * 2u;                                                               \
    }
    // piece I of the stage G8P_STAGE prepared
#define G8P_FIRE(I)                                                                                                           \
    {                                                                                                                         \
        char* dump_ = smem + NS * STAGE;                                                                                      \
        if ((I) < NPA) {                                                                                                      \
            unsigned vo_ = aoff[(I) < NPA ? (I) : 0] + st_ka;                                                                 \
            if (CONV) vo_ = ((amask[(I) < NPA ? (I) : 0] >> st_tap) & 1u) ? vo_ : 0xffffffffu;                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(st_ex ? st_base + (wid + 8 * (I)) * 1024 : dump_), 16, vo_, 0, 0, 0); \
        } else {                                                                                                              \
            const int iw_ = (I) >= NPA ? (I) - NPA : 0;                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)((st_ex && wdst[iw_] >= 0) ? st_base + wdst[iw_] : dump_), 16, woff[iw_] + st_kw, 0, 0, 0); \
        }                                                                                                                     \
    }

    float16v acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    half8 fa[2][MT], fb[2][NT];
#define G8P_READK(KT, KS)                                                                                                     \
    {                                                                                                                         \
        const char* ab_ = smem + ((KT) % NS) * STAGE + (wm * MT * 32) * ROWB;                                                 \
        const char* bb_ = smem + ((KT) % NS) * STAGE + BM * ROWB + (wn * NT * 32) * ROWB;                                     \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) { int R = a * 32 + frow; fa[KS][a] = *(const half8*)(ab_ + R * ROWB + (((2 * (KS) + fh) ^ ((R >> 2) & 3)) << 4)); } \
        _Pragma("unroll") for (int b = 0; b < NT; ++b) { int R = b * 32 + frow; fb[KS][b] = *(const half8*)(bb_ + R * ROWB + (((2 * (KS) + fh) ^ ((R >> 2) & 3)) << 4)); } \
    }
    // the MFMAs [LO, HI) of a step with the pieces of stage KT pinned between them: piece q right after MFMA index q * G of the FIRING order,
    // which is the MFMA order rotated by FOFF (group 1 issues the second part of a stage before the first part of the next one)
#define G8P_MRANGE(KT, LO, HI, FOFF)                                                                                          \
    {                                                                                                                         \
        _Pragma("unroll") for (int mi = (LO); mi < (HI); ++mi) {                                                             \
            const int ks_ = mi / (MT * NT), a_ = (mi / NT) % MT, b_ = mi % NT;                                                \
            acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks_][a_], fb[ks_][b_], acc[a_][b_], 0, 0, 0);            \
            if (!(G8_ABL & 1) && ((mi + (FOFF)) % nM) % G == 0 && ((mi + (FOFF)) % nM) / G < NP) {                            \
                __builtin_amdgcn_sched_barrier(0);                                                                            \
                G8P_FIRE(((mi + (FOFF)) % nM) / G);                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                            \
            }                                                                                                                 \
        }                                                                                                                     \
    }
#ifdef G8_PROF
    unsigned long long tsum_[5] = {0, 0, 0, 0, 0}, tlast_, tnow_;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tlast_));
#define G8P_T(I) { asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tnow_) :: "memory"); tsum_[I] += tnow_ - tlast_; tlast_ = tnow_; }
#else
#define G8P_T(I)
#endif

    // prologue: stages 0 .. 2 whole; stages 0 and 1 landed before the first reads (B_0 certifies stage 1)
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        G8P_STAGE(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) G8P_FIRE(i);
    }
    wait_vm<NP>();
    __builtin_amdgcn_s_barrier();
    G8P_READK(0, 0);
    if (G8_ABL & 2) G8P_READK(0, 1);
#define G8P_READL(KT, KS) { if (!(G8_ABL & 2)) G8P_READK(KT, KS); }
    if (grp == 0) {
        for (int t = 0; t < nk; ++t) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // B_t
            G8P_T(0)
            G8P_READL(t, 1);
            G8P_STAGE(t + PD);
            G8P_MRANGE(t + PD, 0, HALF, 0);
            G8P_T(1)
            G8P_READL(t + 1, 0);
            G8P_MRANGE(t + PD, HALF, nM, 0);
            G8P_T(3)
            wait_vm<NP>();
            G8P_T(4)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        G8P_STAGE(PD);
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (q * G < HALF) G8P_FIRE(q);              // the "half step -1" part of stage 3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // B_0
        for (int t = 0; t < nk; ++t) {
            G8P_READL(t, 1);
            G8P_STAGE(t + PD);
            G8P_MRANGE(t + PD, 0, HALF, HALF);          // second part of stage t+3
            G8P_T(1)
            wait_vm<NP>();
            G8P_T(4)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // B_t+1
            G8P_T(0)
            G8P_READL(t + 1, 0);
            G8P_STAGE(t + 1 + PD);
            G8P_MRANGE(t + 1 + PD, HALF, nM, HALF);     // first part of stage t+4
            G8P_T(3)
        }
    }
#ifdef G8_PROF
    if (g8_prof_buf && lane == 0 && (wid & 3) == 0 && blockIdx.x < 64) {
#pragma unroll
        for (int i = 0; i < 5; ++i) g8_prof_buf[(blockIdx.x * 2 + grp) * 5 + i] = tsum_[i];
    }
#endif
    wait_vm<0>();                                       // dummy pieces (dump area) and group 1's early part of a stage past the end
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the last half step's look-ahead reads (unused) before the epilogue rewrites the LDS
#undef G8P_FIRE
#undef G8P_STAGE
#undef G8P_READK
#undef G8P_READL
#undef G8P_MRANGE
#undef G8P_T
    g8_epilogue<MT, NT, WM, WN>(acc, smem, bias, resid, C, M, N, ldc, ldr, act, m0, n0, wid, lane, wm, wn);
#endif
}

#ifndef G8_LAB_ONLY   // tools/micro/gemm8_lab.hip instantiates single kernels itself
// schedule: 0 = k_gemm8 (round-2 ping-pong, DMA in the LDS-read segment), 1 = k_gemm8s (two barriers, DMA in the MFMA segment), 2 = k_gemm8p
// (half-step pipeline, one barrier per step, buffer-addressed DMA; default).  TCL_GEMM8_SCHED overrides it (experiments; bit-identical results).
// Measured on the metric's shapes (tools/micro/gemm8_lab.hip, profiles/r3_gemm8_lab.txt): dense K >= 640 +17...33 %, 3x3 convs +6...10 % over
// schedule 0; variants that lost and are not in the tree: two early pieces in the read segment (-4 %), static / alternating s_setprio
// (+-1 %), a 24-cycle stagger of the four SIMDs behind each barrier (-8 %).
int g_gemm8_sched = -1;
static int gemm8_sched() {
    if (g_gemm8_sched < 0) { const char* e = getenv("TCL_GEMM8_SCHED"); g_gemm8_sched = e ? atoi(e) : 2; }
    return g_gemm8_sched;
}

template <int MT, int NT, int WM, int WN>
static int launch8(const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K, int lda,
                   int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const int tm = cdiv(M, BM), tn = cdiv(N, BN);
    const size_t ring = (size_t)4 * (BM + BN) * 64 + 1024, epi = (size_t)8 * 32 * (NT * 32 + 8) * 2, lds = ring > epi ? ring : epi;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_gemm8<MT, NT, WM, WN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8<MT, NT, WM, WN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8s<MT, NT, WM, WN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8s<MT, NT, WM, WN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8p<MT, NT, WM, WN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8p<MT, NT, WM, WN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const dim3 grid(cdiv(tm, 8) * 8 * tn);
    int sched = gemm8_sched();
    // k_gemm8p addresses its operands with 32-bit offsets and has no nearest-upsample gather: those calls take k_gemm8s
    const size_t a_bytes = cp.conv ? (size_t)(M / (cp.Hout * cp.Wout)) * cp.Hin * cp.Win * cp.Cin * 2 : ((size_t)(M - 1) * lda + K) * 2;
    const size_t w_bytes = ((size_t)(N - 1) * ldw + K) * 2;
    if (sched == 2 && (a_bytes >= 0xffffff00ull || w_bytes >= 0xffffff00ull || (cp.conv && (cp.Hup != cp.Hin || cp.Wup != cp.Win)))) sched = 1;
#define G8_LAUNCH(KERN, ...) hipLaunchKernelGGL((KERN), grid, dim3(512), lds, st, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, tm, tn, ##__VA_ARGS__)
    if (sched == 2) { if (cp.conv) G8_LAUNCH((k_gemm8p<MT, NT, WM, WN, true>), (unsigned)a_bytes, (unsigned)w_bytes); else G8_LAUNCH((k_gemm8p<MT, NT, WM, WN, false>), (unsigned)a_bytes, (unsigned)w_bytes); }
    else if (sched == 1) { if (cp.conv) G8_LAUNCH((k_gemm8s<MT, NT, WM, WN, true>)); else G8_LAUNCH((k_gemm8s<MT, NT, WM, WN, false>)); }
    else { if (cp.conv) G8_LAUNCH((k_gemm8<MT, NT, WM, WN, true>)); else G8_LAUNCH((k_gemm8<MT, NT, WM, WN, false>)); }
#undef G8_LAUNCH
    return hipPeekAtLastError() == hipSuccess ? TCL_OK : TCL_ELAUNCH;
}

// cfg: 1 = 256x320 (N % 320 == 0), 2 = 128x320, 3 = 256x256 (N % 256 == 0), 4 = 128x256.  Preconditions (checked by the caller):
// K % 64 == 0 (pairs of K steps share 128-B lines; conv: Cin % 64 == 0), N % 8 == 0, ldc % 8 == 0, (ldr % 8 == 0), act in {0, 1};
// act 2 (GEGLU, 64-column [value | gate] groups) on cfg 3 / 4 only.
int gemm8_dispatch(int cfg, const _Float16* A, const _Float16* W, const _Float16* bias, const _Float16* resid, _Float16* C, int M, int N, int K,
                   int lda, int ldw, int ldc, int ldr, int act, const ConvP& cp, hipStream_t st) {
    switch (cfg) {
        case 1: return launch8<2, 5, 4, 2>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
        case 2: return launch8<1, 5, 4, 2>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
        case 3: return launch8<4, 2, 2, 4>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
        case 4: return launch8<2, 2, 2, 4>(A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, st);
    }
    return TCL_EINVAL;
}
#endif
