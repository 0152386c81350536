// Lab: the "256^2 8-phase" GEMM template of /opt/skills/guides/cdna_hip_programming.md:612-660 written out for f16, timed
// beside the product's k_gemm8 / k_gemm8p on the same uniform-random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-comment tools/micro/gemm8ph_lab.hip -o tools/micro/bin/gemm8ph_lab
// Template as written: 256 x 256 tile, BK = 64, 8 waves as 2 (M) x 4 (N), mfma_f32_16x16x32_f16, LDS = 2 buffers x {A, B} x 2 half-tiles of
// [128 rows][64 k] (128-B rows, 16 KiB each, 128 KiB), st_16x32 swizzle on the DMA source address and on the ds_read address, one half-tile
// (2 LDS-DMA instructions per wave) staged per phase, vmcnt(6) at phases 4 and 8 only, two raw barriers per phase, the two wave groups
// (wr = 0 / 1: the two waves of every SIMD) one barrier apart.
//
// A wave (wr, wc) owns the four 64 x 32 quadrants (h, h') of its 128 x 64 output: rows h*128 + wr*64 .. +64, columns h'*128 + wc*32 .. +32 --
// so in every phase ALL waves read the same half-tiles and a half-tile is dead as soon as its phase is over:
//   phase 1: read B0 (4 x b128, retired by lgkmcnt(8) BEFORE the barrier), A0 (8)   -> quadrant (0,0)     stage A1(t+1)
//   phase 2: read B1 (4)                                                            -> (0,1)              stage B0(t+2)   [B0 dead: 1 phase + lgkmcnt rule]
//   phase 3: read A1 (8)                                                            -> (1,1)              stage A0(t+2)   [A0 last read in phase 1]
//   phase 4: -- (A1, B0 still in registers)                                         -> (1,0)              stage B1(t+2)   [B1 last read in phase 2]; vmcnt(6)
// Tile t lives in buffer t & 1; phases 5-8 are phases 1-4 of tile t+1.  At phase 4's vmcnt(6) the three youngest stages (B0, A0, B1 of tile t+2) stay
// in flight and everything older -- all of tile t+1 -- has landed; it is read from phase 5 on, i.e. after the barrier that follows every wave's wait.
// Ticks (barrier intervals; group 0 runs phase p's read part in tick 2(p-1), its MFMA part in tick 2(p-1)+1; group 1 one tick later):
//   WAR  B0(t): last read G1 tick 1, retired (lgkmcnt(8)) before the barrier ending tick 1; restaged by G0 in tick 2.
//        A0(t): last read G1 tick 1, retired (lgkmcnt(0)) in tick 2; restaged G0 tick 4.  B1(t): G1 tick 3 / G0 tick 6.  A1(t): G1 tick 5 / G0 tick 8.
//   RAW  G1's phase-4 wait sits before the barrier ending tick 7; the first read of tile t+1 is G0's in tick 8.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#ifndef PH_ABL
#define PH_ABL 0      // knock-outs (timing only): 1 = no DMA in the loop, 2 = no ds_reads in the loop
#endif

#define PH_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// SWZ 0: st_16x32 as the guide writes it (chunk ^= ((row >> 2) & 1) << 1: 4-way instead of 8-way conflicts); 1: chunk ^= (row >> 1) & 7 (conflict-free
// for 16 consecutive rows of 128 B)
template <int SWZ, int PRIO, int GM = 1>
__global__ __launch_bounds__(512) void k_g8ph(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C, int M, int N, int K,
                                              int lda, int ldw, int ldc, int tiles_m, int tiles_n, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int HALF = 16384, OPB = 32768, BUF = 65536;
    const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    // bijective XCD remap (guide, "XCD swizzle must be bijective")
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    // GM > 1: an XCD's run of tiles walks GM-row bands column by column (its 32 concurrent tiles share GM A panels and 32 / GM B panels)
    const int tm = GM > 1 ? (wg / (GM * tiles_n)) * GM + (wg % (GM * tiles_n)) % GM : wg / tiles_n, tn = GM > 1 ? (wg % (GM * tiles_n)) / GM : wg % tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 2, wc = wid & 3;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);
    // staging: thread -> LDS position (row = tid >> 3 (+64 in the second pass), chunk' = tid & 7) of a half-tile; its SOURCE chunk is chunk' ^ swz(row)
    const int srow = tid >> 3;
    const int sswz = SWZ ? ((srow >> 1) & 7) : (((srow >> 2) & 1) << 1);
    const unsigned a_src = (unsigned)(m0 + srow) * (unsigned)lda * 2u + (unsigned)(((tid & 7) ^ sswz) * 16);
    const unsigned w_src = (unsigned)(n0 + srow) * (unsigned)ldw * 2u + (unsigned)(((tid & 7) ^ sswz) * 16);
    const unsigned a_p1 = 64u * lda * 2u, a_h = 128u * lda * 2u, w_p1 = 64u * ldw * 2u, w_h = 128u * ldw * 2u;
    char* const sdst = smem + wid * 1024;                                       // + buffer + operand + half + pass * 8192
    // fragment reads: lane -> row (lane & 15) of a 16-row block, k-chunk (lane >> 4) of a 32-k block
    const int frow = lane & 15, fc = lane >> 4;
    const int fs = SWZ ? ((frow >> 1) & 7) : (((frow >> 2) & 1) << 1);
    const int foff0 = frow * 128 + ((fc ^ fs) << 4), foff1 = frow * 128 + (((4 + fc) ^ fs) << 4);
    const char* const a_rd = smem + wr * 64 * 128;                               // + buffer + half h
    const char* const b_rd = smem + OPB + wc * 32 * 128;                         // + buffer + half h'

    float4v acc[2][2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][g][i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    half8 fa[4][2], fb0[2][2], fb1[2][2];
    const int nt = K / 64;

    // one half-tile (operand OP, half H) of tile U into buffer U & 1: two LDS-DMA instructions per wave
#define PH_STAGE(OP, H, U)                                                                                                    \
    if (!(PH_ABL & 1)) {                                                                                                      \
        char* d_ = sdst + ((U) & 1) * BUF + (OP) * OPB + (H) * HALF;                                                          \
        const unsigned k_ = (unsigned)(U) * 128u;                                                                             \
        if ((OP) == 0) {                                                                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, PH_LDS(d_), 16, a_src, k_ + (H) * a_h, 0, 0);                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, PH_LDS(d_ + 8192), 16, a_src, k_ + (H) * a_h + a_p1, 0, 0);          \
        } else {                                                                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, PH_LDS(d_), 16, w_src, k_ + (H) * w_h, 0, 0);                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, PH_LDS(d_ + 8192), 16, w_src, k_ + (H) * w_h + w_p1, 0, 0);          \
        }                                                                                                                     \
    }
#define PH_READ_A(H, T)                                                                                                       \
    if (!(PH_ABL & 2)) {                                                                                                      \
        const char* p_ = a_rd + ((T) & 1) * BUF + (H) * HALF;                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { fa[i][0] = *(const half8*)(p_ + i * 2048 + foff0); fa[i][1] = *(const half8*)(p_ + i * 2048 + (SWZ ? foff1 : foff0 + 64)); } \
    }
#define PH_READ_B(FB, H, T)                                                                                                   \
    if (!(PH_ABL & 2)) {                                                                                                      \
        const char* p_ = b_rd + ((T) & 1) * BUF + (H) * HALF;                                                                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { FB[j][0] = *(const half8*)(p_ + j * 2048 + foff0); FB[j][1] = *(const half8*)(p_ + j * 2048 + (SWZ ? foff1 : foff0 + 64)); } \
    }
#define PH_MFMA(H, G, FB)                                                                                                     \
    {                                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
                    acc[H][G][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][kk], FB[j][kk], acc[H][G][i][j], 0, 0, 0); \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
    }
#define PH_BAR() __builtin_amdgcn_s_barrier()
    // the four phases of tile T (buffer T & 1)
#define PH_TILE(T)                                                                                                            \
    {                                                                                                                         \
        /* phase 1 */                                                                                                         \
        PH_READ_B(fb0, 0, T);                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        PH_READ_A(0, T);                                                                                                      \
        if ((T) + 1 < nt) PH_STAGE(0, 1, (T) + 1);                                                                            \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                                    \
        PH_BAR();                                                                                                             \
        PH_MFMA(0, 0, fb0);                                                                                                   \
        PH_BAR();                                                                                                             \
        /* phase 2 */                                                                                                         \
        PH_READ_B(fb1, 1, T);                                                                                                 \
        if ((T) + 2 < nt) PH_STAGE(1, 0, (T) + 2);                                                                            \
        PH_BAR();                                                                                                             \
        PH_MFMA(0, 1, fb1);                                                                                                   \
        PH_BAR();                                                                                                             \
        /* phase 3 */                                                                                                         \
        PH_READ_A(1, T);                                                                                                      \
        if ((T) + 2 < nt) PH_STAGE(0, 0, (T) + 2);                                                                            \
        PH_BAR();                                                                                                             \
        PH_MFMA(1, 1, fb1);                                                                                                   \
        PH_BAR();                                                                                                             \
        /* phase 4 */                                                                                                         \
        if ((T) + 2 < nt) { PH_STAGE(1, 1, (T) + 2); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }                       \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
        PH_BAR();                                                                                                             \
        PH_MFMA(1, 0, fb0);                                                                                                   \
        PH_BAR();                                                                                                             \
    }

    // prologue: tile 0 whole (B0, A0, B1, A1), then the first three half-tiles of tile 1
    PH_STAGE(1, 0, 0); PH_STAGE(0, 0, 0); PH_STAGE(1, 1, 0); PH_STAGE(0, 1, 0);
    if (nt > 1) { PH_STAGE(1, 0, 1); PH_STAGE(0, 0, 1); PH_STAGE(1, 1, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PH_BAR();
    if (wr == 1) PH_BAR();                       // group 1 runs one barrier behind
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        PH_TILE(t);
        PH_TILE(t + 1);
    }
    if (t < nt) PH_TILE(t);
    if (wr == 0) PH_BAR();

    // epilogue (lab): straight from the accumulators; C/D layout of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + h * 128 + wr * 64 + i * 16 + (lane >> 4) * 4 + r, n = n0 + g * 128 + wc * 32 + j * 16 + (lane & 15);
                        C[(long)m * ldc + n] = (_Float16)acc[h][g][i][j][r];
                    }
#endif
}

// ---- the product's 8-wave kernels, for the same-run comparison
#include "../../tc_light_amd/csrc/gemm8.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.f - 1.f) * scale);
    }
}
__global__ void k_ref(const _Float16* A, const _Float16* W, const _Float16* C, int M, int N, int K, int nsamp, float* err) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamp) return;
    unsigned h = s * 747796405u + 2891336453u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const int m = h % (unsigned)M; h = h * 1664525u + 1013904223u; const int n = (h >> 8) % (unsigned)N;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
    err[s] = fabsf((float)C[(size_t)m * N + n] - acc) / (fabsf(acc) + 1.f);
}
// full comparison of two results (different MFMA shapes sum in different orders: not bit-identical) -> max |a - b| / (|b| + 1)
__global__ void k_maxdiff(const _Float16* a, const _Float16* b, size_t n, unsigned* out) {
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = (float)a[i], y = (float)b[i];
        const float d = (x == x) ? fabsf(x - y) / (fabsf(y) + 1.f) : 1e9f;
        mx = fmaxf(mx, d);
    }
    atomicMax(out, __float_as_uint(mx));
}

template <int SWZ, int PRIO, int GM = 1>
static void run_ph(const _Float16* A, const _Float16* W, _Float16* C, int M, int N, int K, hipStream_t st) {
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)k_g8ph<SWZ, PRIO, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); set = true; }
    const int tm = M / 256, tn = N / 256;
    hipLaunchKernelGGL((k_g8ph<SWZ, PRIO, GM>), dim3(tm * tn), dim3(512), 131072, st, A, W, C, M, N, K, K, K, N, tm, tn, (unsigned)((size_t)M * K * 2), (unsigned)((size_t)N * K * 2));
}

int main(int argc, char** argv) {
    struct S { int M, N, K; };
    std::vector<S> shapes = {{4096, 4096, 4096}, {8192, 8192, 8192}, {16384, 1280, 5120}, {57600 / 256 * 256, 1280, 11520}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* derr; CK(hipMalloc(&derr, 8192 * 4));
    unsigned* dmx; CK(hipMalloc(&dmx, 4));
    for (size_t si = 0; si < shapes.size(); ++si) {
        if (only >= 0 && (int)si != only) continue;
        const int M = shapes[si].M, N = shapes[si].N, K = shapes[si].K;
        _Float16 *A, *W, *C0, *C1;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C0, (size_t)M * N * 2)); CK(hipMalloc(&C1, (size_t)M * N * 2));
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, A, (size_t)M * K, 0x1234u + (unsigned)si, 1.0f);
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, W, (size_t)N * K, 0x9876u + (unsigned)si, 1.0f);       // uniform [-1, 1) both, like the guide's bench
        CK(hipStreamSynchronize(st));
        const double flop = 2.0 * M * N * K;
        printf("== %d x %d x %d  (%.1f GFLOP), uniform random [-1,1) operands\n", M, N, K, flop / 1e9);
        ConvP cp = {};
        // variants: 0 = k_gemm8 256x256 (round-2 ping-pong), 1 = k_gemm8p 256x256 (product default), 2 = 8-phase st_16x32 + setprio (as written),
        //           3 = 8-phase st_16x32 without setprio, 4 = 8-phase with the conflict-free swizzle + setprio
        const char* names[7] = {"k_gemm8  256x256 (r2 ping-pong)", "k_gemm8p 256x256 (product)", "8-phase st_16x32 setprio (guide)", "8-phase st_16x32 no setprio", "8-phase full swizzle setprio", "8-phase st_16x32, 4-row bands", "8-phase st_16x32, 8-row bands"};
        auto launch = [&](int v, _Float16* C) {
            if (v == 0) { g_gemm8_sched = 0; gemm8_dispatch(3, A, W, nullptr, nullptr, C, M, N, K, K, K, N, N, 0, cp, st); }
            else if (v == 1) { g_gemm8_sched = 2; gemm8_dispatch(3, A, W, nullptr, nullptr, C, M, N, K, K, K, N, N, 0, cp, st); }
            else if (v == 2) run_ph<0, 1>(A, W, C, M, N, K, st);
            else if (v == 3) run_ph<0, 0>(A, W, C, M, N, K, st);
            else if (v == 4) run_ph<1, 1>(A, W, C, M, N, K, st);
            else if (v == 5) run_ph<0, 0, 4>(A, W, C, M, N, K, st);
            else run_ph<0, 0, 8>(A, W, C, M, N, K, st);
        };
        const int NV = (M / 256) % 8 == 0 ? 7 : 5;
        std::vector<float> tms[7];
        for (int v = 0; v < NV; ++v) {
            _Float16* C = v == 1 ? C0 : C1;
            CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 2, st));
            launch(v, C);
            CK(hipStreamSynchronize(st));
            hipLaunchKernelGGL(k_ref, dim3(32), dim3(256), 0, st, A, W, C, M, N, K, 8192, derr);
            std::vector<float> herr(8192);
            CK(hipMemcpyAsync(herr.data(), derr, 8192 * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            float mx = 0; for (float x : herr) mx = std::max(mx, x);
            unsigned md = 0;
            if (v >= 2 && !PH_ABL) {
                for (int rep = 0; rep < 3; ++rep) {                                  // race screen: repeated launches must agree with the product result everywhere
                    launch(v, C1);
                    hipLaunchKernelGGL(k_maxdiff, dim3(2048), dim3(256), 0, st, C1, C0, (size_t)M * N, dmx);
                }
                CK(hipMemcpyAsync(&md, dmx, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
                CK(hipMemsetAsync(dmx, 0, 4, st));
            }
            float mdf; __builtin_memcpy(&mdf, &md, 4);
            printf("   check %-34s max rel err vs f32 reference (8192 samples) %.2e%s\n", names[v], mx, v >= 2 ? (std::string("   max |x - product| / (|product| + 1) over all outputs, 3 launches: ") + std::to_string(mdf)).c_str() : "");
        }
        for (int round = 0; round < 7; ++round)
            for (int v = 0; v < NV; ++v) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < 4; ++r) launch(v, C1);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                tms[v].push_back(ms / 4);
            }
        for (int v = 0; v < NV; ++v) {
            std::sort(tms[v].begin(), tms[v].end());
            printf("   %-34s median %8.1f us  %6.0f TFLOP/s   (best %6.0f)\n", names[v], tms[v][3] * 1e3, flop / tms[v][3] / 1e9, flop / tms[v][0] / 1e9);
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C0)); CK(hipFree(C1));
    }
    return 0;
}
