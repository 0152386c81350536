// Issue cost (shader cycles per wave64 instruction) of the vector instructions in the flash softmax, measured in-kernel with s_memtime:
// one block of 256 * W threads per CU, W waves per SIMD (W = 1, 2, 4), each wave a straight run of 64 independent instructions x 256 laps.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
template <int OP>
__global__ void k(unsigned long long* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    half8 hv = {1, 2, 3, 4, 5, 6, 7, 8};
    float16v acc = {0}; float4v acc4 = {0};
    float16v accb = {0}; float4v acc4b = {0};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int lap = 0; lap < 256; ++lap) {
        if (OP == 0) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 2) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 3) { REP8(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %0, %3\n v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (OP == 4) { REP8(asm volatile("v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1\n v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (OP == 5) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 6) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(*(double*)&a0), "+v"(*(double*)&a2));) }
        if (OP == 7) { for (int i = 0; i < 8; ++i) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accb, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accb, 0, 0, 0);
                                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accb, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accb, 0, 0, 0); } }
        if (OP == 8) { for (int i = 0; i < 8; ++i) { acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4, 0, 0, 0); acc4b = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4b, 0, 0, 0); acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4, 0, 0, 0); acc4b = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4b, 0, 0, 0);
                                            acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4, 0, 0, 0); acc4b = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4b, 0, 0, 0); acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4, 0, 0, 0); acc4b = __builtin_amdgcn_mfma_f32_16x16x32_f16(hv, hv, acc4b, 0, 0, 0); } }
        // round 6: the K = 8 / K = 16 forms of CDNA3, still in the gfx950 ISA: does a 32x32x8 issue in HALF the cycles of 32x32x16?  If so QK^T at
        // head_dim 40 could run as 2 x K=16 + 1 x K=8 (40 columns) instead of 3 x K=16 (48: 17 % of its matrix cycles are padding)
        if (OP == 10) { half4 h4 = {1, 2, 3, 4}; for (int i = 0; i < 32; ++i) { acc = __builtin_amdgcn_mfma_f32_32x32x8f16(h4, h4, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_32x32x8f16(h4, h4, accb, 0, 0, 0); } }
        if (OP == 11) { half4 h4 = {1, 2, 3, 4}; for (int i = 0; i < 32; ++i) { acc4 = __builtin_amdgcn_mfma_f32_16x16x16f16(h4, h4, acc4, 0, 0, 0); acc4b = __builtin_amdgcn_mfma_f32_16x16x16f16(h4, h4, acc4b, 0, 0, 0); } }
        if (OP == 9) {   // the flash mix per MFMA: 32x32x16 MFMA + 4 exp + 2 cvt, all independent
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, acc, 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accb, 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } }
    }
    const unsigned long long t1 = clock64();
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3) + acc[0] + acc4[0] + accb[1] + acc4b[1];
    // the arbiter favours the oldest wave: the block's time is that of its LAST wave
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMax(out, t1 - t0); out[1] = (unsigned long long)sink; }
}
template <int OP> void run(const char* name, int ninst, unsigned long long* d) {
    for (int w = 1; w <= 4; w *= 2) {      // waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * w), 0, 0, d, 1.0f);
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * w), 0, 0, d, 1.0f);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-28s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, %6.2f per SIMD\n", name, w, (double)h[0] / (256.0 * ninst), (double)h[0] / (256.0 * ninst * w));
    }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    run<0>("v_exp_f32", 64, d); run<1>("v_mul_f32", 64, d); run<2>("v_cvt_pk_f16_f32", 64, d); run<3>("v_permlane16_swap_b32", 64, d);
    run<4>("v_or3_b32", 64, d); run<5>("v_max3_f32", 64, d); run<6>("v_pk_mul_f32", 64, d);
    run<7>("mfma_32x32x16_f16", 64, d); run<8>("mfma_16x16x32_f16", 64, d); run<9>("mix: mfma32 + 4 exp + 2 cvt", 8 * 2, d);
    run<10>("mfma_32x32x8_f16", 64, d); run<11>("mfma_16x16x16_f16", 64, d);
    return 0;
}
