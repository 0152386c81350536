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
template <int MT, int NT, int WM, int WN, bool CONV, int NL>
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
    constexpr int nM = 2 * MT * NT, PM = NP - NL, G = nM / (PM > 0 ? PM : 1);
    static_assert(A_P % 8 == 0 && NL >= 0 && NL <= NP && G >= 1, "tile / schedule");
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
    // M(t) with the PM late pieces of stage KT pinned between the MFMAs: piece NL + q right after MFMA q * G
#define G8S_MSEG(KT)                                                                                                          \
    {                                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                                        \
        _Pragma("unroll") for (int mi = 0; mi < nM; ++mi) {                                                                  \
            const int ks_ = mi / (MT * NT), a_ = (mi / NT) % MT, b_ = mi % NT;                                                \
            acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks_][a_], fb[ks_][b_], acc[a_][b_], 0, 0, 0);            \
            if (PM > 0 && mi % G == 0 && mi / G < PM) {                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                                            \
                G8S_FIRE(KT, NL + mi / G);                                                                                    \
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
    if (grp == 0) {
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            G8S_READ(t);
            G8S_PREP(t + PD);
#pragma unroll
            for (int i = 0; i < NL; ++i) G8S_FIRE(t + PD, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            G8S_MSEG(t + PD);
            G8S_WAIT0();
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            G8S_READ(t);
            G8S_PREP(t + PD);
#pragma unroll
            for (int i = 0; i < NL; ++i) G8S_FIRE(t + PD, i);
            G8S_WAIT1();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            G8S_MSEG(t + PD);
        }
    }
    wait_vm<0>();                                       // dummy pieces of the last steps (dump area) before the epilogue reuses the LDS
#undef G8S_GLDS
#undef G8S_PREP
#undef G8S_FIRE
#undef G8S_READ
#undef G8S_MSEG
#undef G8S_WAIT0
#undef G8S_WAIT1
    g8_epilogue<MT, NT, WM, WN>(acc, smem, bias, resid, C, M, N, ldc, ldr, act, m0, n0, wid, lane, wm, wn);
}

#ifndef G8_LAB_ONLY   // tools/micro/gemm8_lab.hip instantiates single kernels itself
// schedule: 0 = k_gemm8 (round-2 ping-pong, DMA in the LDS-read segment), 1 = k_gemm8s with every piece in the MFMA segment, 2 = k_gemm8s with
// the first two pieces of a stage still in the read segment.  TCL_GEMM8_SCHED overrides the default (experiments; results are bit-identical).
int g_gemm8_sched = -1;
static int gemm8_sched() {
    if (g_gemm8_sched < 0) { const char* e = getenv("TCL_GEMM8_SCHED"); g_gemm8_sched = e ? atoi(e) : 1; }
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
        (void)hipFuncSetAttribute((const void*)k_gemm8s<MT, NT, WM, WN, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8s<MT, NT, WM, WN, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8s<MT, NT, WM, WN, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_gemm8s<MT, NT, WM, WN, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const dim3 grid(cdiv(tm, 8) * 8 * tn);
    const int sched = gemm8_sched();
#define G8_LAUNCH(KERN) hipLaunchKernelGGL((KERN), grid, dim3(512), lds, st, A, W, bias, resid, C, M, N, K, lda, ldw, ldc, ldr, act, cp, tm, tn)
    if (sched == 1) { if (cp.conv) G8_LAUNCH((k_gemm8s<MT, NT, WM, WN, true, 0>)); else G8_LAUNCH((k_gemm8s<MT, NT, WM, WN, false, 0>)); }
    else if (sched == 2) { if (cp.conv) G8_LAUNCH((k_gemm8s<MT, NT, WM, WN, true, 2>)); else G8_LAUNCH((k_gemm8s<MT, NT, WM, WN, false, 2>)); }
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
