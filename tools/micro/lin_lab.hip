// Stand-alone lab for the strip-resident K = 320 Linear (csrc/linstrip.hip, cfg 12) against the tiled kernels of csrc/gemm.hip (no torch).
//   python -c "import __graft_entry__ as g; g.build()"       (objects under tc_light_amd/csrc/build/)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-comment -c tools/micro/lin_lab.hip -o scratch/lin_lab.o
//   hipcc --offload-arch=gfx950 scratch/lin_lab.o tc_light_amd/csrc/build/{gemm,gemm8,linstrip}.o -o tools/micro/bin/lin_lab
// For every shape: cfg 12 is compared bit for bit with cfg 1 (k_gemm_dma 128x128) and timed beside cfgs 1, 3, 11 (median of 5 x 4 launches).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern "C" {
int tcl_gemm_f16(const void* A, const void* W, const void* bias, const void* resid, void* C, int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act, hipStream_t st);
int tcl_gemm_tune(int cfg, int splits);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.f - 1.f) * scale);
    }
}
__global__ void k_diff(const unsigned short* a, const unsigned short* b, int M, int N, int ldc, unsigned* cnt) {
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * N; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / N, n = i % N;
        c += a[m * ldc + n] != b[m * ldc + n];
    }
    if (c) atomicAdd(cnt, c);
}
struct Shape { int M, N, K, act, resid, ldc; const char* name; };

int main(int argc, char** argv) {
    std::vector<Shape> shapes = {
        {368640, 320, 320, 0, 0, 320, "pin / q2 (level 0)"},
        {368640, 320, 320, 0, 1, 320, "o2 / pout (+resid)"},
        {368640, 960, 320, 0, 0, 960, "qkv batched"},
        {95040, 960, 320, 0, 0, 960, "qkv of one merged chunk"},
        {95040, 320, 320, 0, 0, 320, "o1 of one merged chunk"},
        {368640, 2560, 320, 2, 0, 1280, "ff1 GEGLU"},
        {1474560, 320, 320, 0, 1, 320, "o2 / pout, 1.5 M rows"},
        {1474560, 2560, 320, 2, 0, 1280, "ff1 GEGLU, 1.5 M rows"},
        {100001, 352, 320, 1, 1, 360, "odd M, N % 128 != 0, SiLU + resid, ldc 360"},
        {20000, 192, 320, 5, 1, 192, "small, act 5 (GELU after resid)"},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* dcnt; CK(hipMalloc(&dcnt, 4));
    for (size_t si = 0; si < shapes.size(); ++si) {
        if (only >= 0 && (int)si != only) continue;
        const Shape s = shapes[si];
        const int No = s.act == 2 ? s.N / 2 : s.N;
        _Float16 *A, *W, *B, *R, *C0, *C1;
        CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&W, (size_t)s.N * s.K * 2)); CK(hipMalloc(&B, s.N * 2));
        CK(hipMalloc(&R, (size_t)s.M * s.ldc * 2)); CK(hipMalloc(&C0, (size_t)s.M * s.ldc * 2)); CK(hipMalloc(&C1, (size_t)s.M * s.ldc * 2));
        k_fill<<<2048, 256, 0, st>>>(A, (size_t)s.M * s.K, 1u, 1.f);
        k_fill<<<256, 256, 0, st>>>(W, (size_t)s.N * s.K, 2u, 0.08f);
        k_fill<<<16, 256, 0, st>>>(B, s.N, 3u, 0.5f);
        k_fill<<<2048, 256, 0, st>>>(R, (size_t)s.M * s.ldc, 4u, 1.f);
        CK(hipMemsetAsync(C0, 0, (size_t)s.M * s.ldc * 2, st)); CK(hipMemsetAsync(C1, 0xff, (size_t)s.M * s.ldc * 2, st));
        const double gflop = 2.0 * s.M * s.N * s.K / 1e9, gbytes = 2.0 * ((double)s.M * s.K + (double)s.N * s.K + (double)s.M * No * (s.resid ? 2 : 1)) / 1e9;
        printf("== %s: M=%d N=%d K=%d act=%d resid=%d  %.1f GFLOP  %.3f GB\n", s.name, s.M, s.N, s.K, s.act, s.resid, gflop, gbytes);
        auto run = [&](int cfg, _Float16* C) {
            tcl_gemm_tune(cfg, 1);
            return tcl_gemm_f16(A, W, B, s.resid ? R : nullptr, C, s.M, s.N, s.K, s.K, s.K, s.ldc, s.ldc, s.act, st);
        };
        if (run(1, C0) != 0) { printf("   cfg 1 refused\n"); continue; }
        const int rc = run(12, C1);
        CK(hipStreamSynchronize(st));
        if (rc != 0) { printf("   cfg 12 refused (rc %d)\n", rc); }
        else {
            CK(hipMemsetAsync(dcnt, 0, 4, st));
            k_diff<<<1024, 256, 0, st>>>((const unsigned short*)C0, (const unsigned short*)C1, s.M, No, s.ldc, dcnt);
            unsigned h = 0; CK(hipMemcpyAsync(&h, dcnt, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            printf("   cfg 12 vs cfg 1: %u of %zu outputs differ %s\n", h, (size_t)s.M * No, h ? "<-- MISMATCH" : "(bit-identical)");
        }
        const int cfgs[] = {1, 3, 11, 6, 7, 12};
        for (int cfg : cfgs) {
            if (cfg == 6 && (s.N % 320 || s.act == 2)) continue;
            if (cfg == 7 && s.N % 256) continue;
            if (run(cfg, C1) != 0) continue;
            std::vector<float> t;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 4; ++i) run(cfg, C1);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 4 * 1e3f);
            }
            std::sort(t.begin(), t.end());
            printf("   cfg %2d: %8.1f us  %6.0f TF  %5.2f TB/s  (best %.1f us)\n", cfg, t[2], gflop / t[2] * 1e3, gbytes / t[2] * 1e3, t[0]);
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(B)); CK(hipFree(R)); CK(hipFree(C0)); CK(hipFree(C1));
    }
    return 0;
}
