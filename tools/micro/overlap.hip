// How well does one SIMD overlap matrix and vector work?  Two (or four) waves per SIMD, wave roles by index, time of the LAST wave.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
#define EXP8 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define MUL8 asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define CVT8 asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define MF4 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, acc, 0, 0, 0); accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accb, 0, 0, 0); accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accc, 0, 0, 0); accd = __builtin_amdgcn_mfma_f32_32x32x16_f16(hv, hv, accd, 0, 0, 0);
// MODE 0: every wave: 256 MFMA.               MODE 1: every wave: 1024 exp.         MODE 2: every wave: 2048 mul
// MODE 3: even waves 256 MFMA, odd waves 1024 exp (roles split across the waves of a SIMD: waves w and w+4 share a SIMD)
// MODE 4: even MFMA, odd 2048 mul             MODE 5: even MFMA, odd 2048 cvt
// MODE 6: every wave alternates [4 MFMA][16 exp] (fine-grained, same phase)      MODE 7: [16 MFMA][64 exp] coarse, waves on a SIMD start in opposite phases
// MODE 8: every wave alternates [4 MFMA][32 mul]
template <int MODE>
__global__ void k(unsigned long long* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    half8 hv = {1, 2, 3, 4, 5, 6, 7, 8};
    float16v acc = {0}, accb = {0}, accc = {0}, accd = {0};
    const int wid = threadIdx.x >> 6, role = (wid >> 2) & 1;       // waves w, w+4 (, w+8, w+12) share SIMD w & 3
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int lap = 0; lap < 64; ++lap) {
        if (MODE == 0 || ((MODE == 3 || MODE == 4 || MODE == 5) && role == 0)) { MF4 MF4 MF4 MF4 }
        if (MODE == 1 || (MODE == 3 && role == 1)) { EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 }
        if (MODE == 2 || (MODE == 4 && role == 1)) { MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 MUL8 }
        if (MODE == 5 && role == 1) { CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 CVT8 }
        if (MODE == 6) { MF4 EXP8 EXP8 MF4 EXP8 EXP8 MF4 EXP8 EXP8 MF4 EXP8 EXP8 }
        if (MODE == 7) { if (role == 0) { MF4 MF4 MF4 MF4 EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 } else { EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 EXP8 MF4 MF4 MF4 MF4 } }
        if (MODE == 8) { MF4 MUL8 MUL8 MUL8 MUL8 MF4 MUL8 MUL8 MUL8 MUL8 MF4 MUL8 MUL8 MUL8 MUL8 MF4 MUL8 MUL8 MUL8 MUL8 }
    }
    const unsigned long long t1 = clock64();
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc[0] + accb[1] + accc[2] + accd[3];
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMax(out, t1 - t0); out[1] = (unsigned long long)sink; }
}
template <int MODE> void run(const char* name, unsigned long long* d) {
    for (int w = 1; w <= 4; w *= 2) {
        if (w == 1 && (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7)) continue;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * w), 0, 0, d, 1.0f);
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * w), 0, 0, d, 1.0f);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-64s %d wave(s)/SIMD: %8.0f cycles per lap (last wave)\n", name, w, (double)h[0] / 64.0);
    }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    run<0>("all waves: 16 MFMA 32x32x16 (512 pipe cycles)", d);
    run<1>("all waves: 64 v_exp", d);
    run<2>("all waves: 128 v_mul", d);
    run<3>("even waves 16 MFMA | odd waves 64 v_exp", d);
    run<4>("even waves 16 MFMA | odd waves 128 v_mul", d);
    run<5>("even waves 16 MFMA | odd waves 128 v_cvt_pk", d);
    run<6>("all waves: 4 x [4 MFMA, 16 exp]", d);
    run<7>("role 0: [16 MFMA][64 exp]; role 1: [64 exp][16 MFMA]", d);
    run<8>("all waves: 4 x [4 MFMA, 32 mul]", d);
    return 0;
}
