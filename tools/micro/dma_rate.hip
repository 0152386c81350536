// How fast can one CU move L2-resident operand tiles into LDS?  (stand-alone: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_rate.hip)
// 256 blocks x 512 threads (8 waves, one block per CU).  Every wave moves NIT pieces of 1 KiB from a per-block window of a row-major
// f16 matrix (row stride LD bytes) into a 128 KiB LDS ring; modes:
//   0  LDS-DMA, piece = 16 rows x 64 B   (k_gemm8's K = 32 stages)
//   1  LDS-DMA, piece =  8 rows x 128 B  (full cache lines, K = 64 stages)
//   2  LDS-DMA, every lane reads the same 16 B (no fetch traffic at all: the LDS-write side alone)
//   3  LDS-DMA through a buffer descriptor, 16 rows x 64 B
//   4  register staging: global_load_dwordx4 -> ds_write_b128, 16 rows x 64 B
//   5  register staging, 8 rows x 128 B
//   6  LDS-DMA, piece = 4 rows x 256 B
// Reported: bytes per shader cycle and CU (s_memtime of wave 0), aggregate TB/s from the launch wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k_dma(const char* __restrict__ src, int ld, int nit, unsigned bytes, unsigned long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x % 64) * 256 * ld;            // 64 windows of 256 rows: L2-resident, shared by 4 blocks each
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    u32x4 stage[4];
    for (int it = 0; it < nit; ++it) {
        // piece (it, wid): rows of the window, 64-byte (or 128 / 256) column block advancing with it
        const int slot = (it & 15) * 8 + wid;                                 // 128 x 1 KiB ring
        char* dst = smem + slot * 1024;
        int row, col;
        if (MODE == 0 || MODE == 3 || MODE == 4) { row = wid * 16 + (lane >> 2) + ((it >> 4) & 1) * 128; col = (it % 40) * 64 + (lane & 3) * 16; }
        else if (MODE == 1 || MODE == 5) { row = wid * 8 + (lane >> 3) + ((it >> 4) & 3) * 64; col = (it % 20) * 128 + (lane & 7) * 16; }
        else if (MODE == 6) { row = wid * 4 + (lane >> 4) + ((it >> 4) & 7) * 32; col = (it % 10) * 256 + (lane & 15) * 16; }
        else { row = 0; col = 0; }
        const char* p = base + (size_t)row * ld + col;
        if (MODE == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (unsigned)(p - src), 0, 0, 0);
        else if (MODE == 4 || MODE == 5) {
            stage[it & 3] = *(const u32x4*)p;
            if (it >= 3) { *(u32x4*)(smem + (((it - 3) & 15) * 8 + wid) * 1024 + lane * 16) = stage[(it - 3) & 3]; }
        } else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        if ((it & 7) == 7 && MODE != 4 && MODE != 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE == 4 || MODE == 5) { if (smem[tid] == 77) cyc[0] = 0; }
#endif
}

template <int MODE> void run(const char* name, const char* src, int ld, unsigned bytes, unsigned long long* dc) {
    const int nit = 4096;
    CK(hipFuncSetAttribute((const void*)k_dma<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_dma<MODE>, dim3(256), dim3(512), 131072, 0, src, ld, nit, bytes, dc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_dma<MODE>, dim3(256), dim3(512), 131072, 0, src, ld, nit, bytes, dc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(256); CK(hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v / 256;
    const double per_cu = (double)nit * 8 * 1024;
    printf("%-58s %6.1f B/clk/CU  (%.0f cycles per KiB piece)  %6.2f TB/s aggregate, clock %.2f GHz\n", name, per_cu / mean, mean / (nit * 8.0), per_cu * 256 / ms / 1e9, mean / ms / 1e6);
}

int main() {
    const int ld = 5760, rows = 64 * 256;
    const unsigned bytes = (unsigned)rows * ld;
    char* src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
    unsigned long long* dc; CK(hipMalloc(&dc, 256 * 8));
    run<0>("LDS-DMA 16 rows x 64 B (global_load_lds)", src, ld, bytes, dc);
    run<1>("LDS-DMA 8 rows x 128 B", src, ld, bytes, dc);
    run<6>("LDS-DMA 4 rows x 256 B", src, ld, bytes, dc);
    run<2>("LDS-DMA same 16 B for every lane", src, ld, bytes, dc);
    run<3>("LDS-DMA 16 rows x 64 B (buffer_load ... lds)", src, ld, bytes, dc);
    run<4>("register staging 16 rows x 64 B (load + ds_write_b128)", src, ld, bytes, dc);
    run<5>("register staging 8 rows x 128 B", src, ld, bytes, dc);
    return 0;
}
