// Stand-alone lab for the 8-wave GEMM / implicit-conv kernels (no Python, no torch: builds here with hipcc, runs on the GPU box in seconds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-comment tools/micro/gemm8_lab.hip -o gpurun_out/gemm8_lab && gpurun_out/gemm8_lab
// For every shape: each valid tile configuration x schedule (0 = round-2 ping-pong, 1 = DMA in the MFMA segment, 2 = half-step pipeline) is
// (a) compared bit for bit with schedule 0 of the same tile, (b) checked on 8192 sampled outputs against an f32 reference kernel,
// (c) timed in interleaved rounds (median and best of 5 x 4 launches).  Output: one line per (shape, cfg, schedule) with TFLOP/s.
#ifdef LAB_PROF
#define G8_PROF
#endif
#include "../../tc_light_amd/csrc/gemm8.hip"
#include "../../tc_light_amd/csrc/gemm8q.hip"
#include <algorithm>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.f - 1.f) * scale);
    }
}
// sampled f32 reference: sample s -> (m, n) by hashing
__global__ void k_ref(const _Float16* A, const _Float16* W, const _Float16* C, int M, int N, int K, ConvP cp, int nsamp, float* err) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamp) return;
    unsigned h = s * 747796405u + 2891336453u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const int m = h % (unsigned)M; h = h * 1664525u + 1013904223u; const int n = (h >> 8) % (unsigned)N;
    float acc = 0.f;
    if (!cp.conv) { for (int k = 0; k < K; ++k) acc += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k]; }
    else {
        const int hw = cp.Hout * cp.Wout, b = m / hw, r = m - b * hw, oy = r / cp.Wout, ox = r - oy * cp.Wout;
        for (int tap = 0; tap < 9; ++tap) {
            int iy = oy * cp.stride - cp.pad + tap / 3, ix = ox * cp.stride - cp.pad + tap % 3;
            if (iy < 0 || iy >= cp.Hup || ix < 0 || ix >= cp.Wup) continue;
            if (cp.Hup != cp.Hin || cp.Wup != cp.Win) { iy = min((int)floorf(iy * cp.sy), cp.Hin - 1); ix = min((int)floorf(ix * cp.sx), cp.Win - 1); }
            const _Float16* xp = A + (((size_t)b * cp.Hin + iy) * cp.Win + ix) * cp.Cin;
            const _Float16* wp = W + (size_t)n * K + tap * cp.Cin;
            for (int c = 0; c < cp.Cin; ++c) acc += (float)xp[c] * (float)wp[c];
        }
    }
    const float got = (float)C[(size_t)m * N + n];
    err[s] = fabsf(got - acc) / (fabsf(acc) + 1.f);
}
__global__ void k_diff(const unsigned* a, const unsigned* b, size_t n, unsigned* cnt) {
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}

struct Shape { int conv, M, N, K, B, H, W, Cin, up; const char* name; int stride = 1, pad = 1; };

int main(int argc, char** argv) {
    std::vector<Shape> shapes = {
        {0, 368640, 320, 320, 0, 0, 0, 0, 0, "lin 368640x320x320 (q2/o2/pin/pout, level 0)"},
        {0, 368640, 320, 1280, 0, 0, 0, 0, 0, "lin 368640x320x1280 (ff2, level 0)"},
        {0, 95040, 960, 320, 0, 0, 0, 0, 0, "lin 95040x960x320 (qkv of one merged chunk)"},
        {0, 92160, 640, 640, 0, 0, 0, 0, 0, "lin 92160x640x640 (level 1)"},
        {0, 92160, 640, 2560, 0, 0, 0, 0, 0, "lin 92160x640x2560 (ff2, level 1)"},
        {0, 23040, 1280, 1280, 0, 0, 0, 0, 0, "lin 23040x1280x1280 (level 2)"},
        {0, 23040, 1280, 5120, 0, 0, 0, 0, 0, "lin 23040x1280x5120 (ff2, level 2)"},
        {0, 8192, 8192, 8192, 0, 0, 0, 0, 0, "lin 8192^3 (guide's reference shape)"},
        {1, 0, 320, 0, 16, 90, 160, 320, 0, "conv 320->320 @90x160 x16"},
        {1, 0, 320, 0, 16, 90, 160, 640, 0, "conv 640->320 @90x160 x16"},
        {1, 0, 320, 0, 16, 90, 160, 960, 0, "conv 960->320 @90x160 x16"},
        {1, 0, 640, 0, 16, 45, 80, 640, 0, "conv 640->640 @45x80 x16"},
        {1, 0, 640, 0, 16, 45, 80, 1280, 0, "conv 1280->640 @45x80 x16"},
        {1, 0, 640, 0, 8, 45, 80, 640, 1, "conv 640->640 up 45x80->90x160 x8"},
        {1, 0, 1280, 0, 16, 23, 40, 1280, 0, "conv 1280->1280 @23x40 x16"},
        {1, 0, 1280, 0, 16, 23, 40, 2560, 0, "conv 2560->1280 @23x40 x16"},
        {1, 0, 1280, 0, 64, 12, 20, 1280, 0, "conv 1280->1280 @12x20 x64"},
        // the metric's pass (300 x 1280 x 720, block-major): up to 1.5 M level-0 rows per launch
        {0, 1474560, 320, 320, 0, 0, 0, 0, 0, "lin 1474560x320x320"},
        {0, 1474560, 320, 1280, 0, 0, 0, 0, 0, "lin 1474560x320x1280 (ff2, level 0)"},
        {0, 368640, 640, 2560, 0, 0, 0, 0, 0, "lin 368640x640x2560 (ff2, level 1)"},
        {0, 92160, 1280, 5120, 0, 0, 0, 0, 0, "lin 92160x1280x5120 (ff2, level 2)"},
        {1, 0, 320, 0, 64, 90, 160, 320, 0, "conv 320->320 @90x160 x64"},
        {1, 0, 320, 0, 64, 90, 160, 640, 0, "conv 640->320 @90x160 x64"},
        {1, 0, 640, 0, 64, 45, 80, 640, 0, "conv 640->640 @45x80 x64"},
        {1, 0, 640, 0, 64, 45, 80, 1280, 0, "conv 1280->640 @45x80 x64"},
        {1, 0, 1280, 0, 64, 23, 40, 1280, 0, "conv 1280->1280 @23x40 x64"},
        {1, 0, 1280, 0, 64, 23, 40, 2560, 0, "conv 2560->1280 @23x40 x64"},
        // stride 2 (UNet down-samplers; the VAE encoder's pad-0 variant with its asymmetric (0,1,0,1) padding): the tap masks of k_gemm8p
        {1, 0, 320, 0, 64, 90, 160, 320, 0, "conv 320->320 stride 2 @90x160 x64", 2, 1},
        {1, 0, 256, 0, 16, 180, 320, 256, 0, "conv 256->256 stride 2 pad 0 @180x320 x16 (VAE encoder)", 2, 0},
        // K sweep at the level-0 Linear size: fixed cost per tile (prologue + epilogue) vs cost per K tile
        {0, 368640, 320, 64, 0, 0, 0, 0, 0, "lin 368640x320x64"}, {0, 368640, 320, 128, 0, 0, 0, 0, 0, "lin 368640x320x128"},
        {0, 368640, 320, 256, 0, 0, 0, 0, 0, "lin 368640x320x256"}, {0, 368640, 320, 640, 0, 0, 0, 0, 0, "lin 368640x320x640"},
        {0, 95040, 960, 64, 0, 0, 0, 0, 0, "lin 95040x960x64"}, {0, 95040, 960, 640, 0, 0, 0, 0, 0, "lin 95040x960x640"},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const unsigned smask = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 0) : 0x7u;      // bit s = time schedule s
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* derr; CK(hipMalloc(&derr, 8192 * 4));
    unsigned* dcnt; CK(hipMalloc(&dcnt, 4));
    for (size_t si = 0; si < shapes.size(); ++si) {
        if (only >= 0 && (int)si != only) continue;
        Shape s = shapes[si];
        ConvP cp = {};
        int M = s.M, N = s.N, K = s.K;
        size_t a_elems;
        if (s.conv) {
            cp.conv = 1; cp.Hin = s.H; cp.Win = s.W; cp.Cin = s.Cin; cp.Hup = s.up ? 2 * s.H : s.H; cp.Wup = s.up ? 2 * s.W : s.W;
            cp.stride = s.stride; cp.pad = s.pad; cp.sy = (float)cp.Hin / cp.Hup; cp.sx = (float)cp.Win / cp.Wup;
            cp.Hout = s.pad ? (cp.Hup + 2 - 3) / s.stride + 1 : (cp.Hup + 1 - 3) / s.stride + 1;
            cp.Wout = s.pad ? (cp.Wup + 2 - 3) / s.stride + 1 : (cp.Wup + 1 - 3) / s.stride + 1;
            M = s.B * cp.Hout * cp.Wout; K = 9 * s.Cin; a_elems = (size_t)s.B * s.H * s.W * s.Cin;
        } else a_elems = (size_t)M * K;
        _Float16 *A, *Wt, *C0, *C1;
        CK(hipMalloc(&A, a_elems * 2)); CK(hipMalloc(&Wt, (size_t)N * K * 2)); CK(hipMalloc(&C0, (size_t)M * N * 2)); CK(hipMalloc(&C1, (size_t)M * N * 2));
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, A, a_elems, 0x1234u + (unsigned)si, 1.0f);
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, Wt, (size_t)N * K, 0x9876u + (unsigned)si, 0.05f);
        CK(hipStreamSynchronize(st));
        const double flop = 2.0 * M * N * K;
        printf("== %s: M=%d N=%d K=%d  %.1f GFLOP\n", s.name, M, N, K, flop / 1e9);
        std::vector<int> cfgs;
        if (N % 320 == 0) { cfgs.push_back(1); cfgs.push_back(2); }
        if (N % 256 == 0) { cfgs.push_back(3); cfgs.push_back(4); }
        const int lda = s.conv ? 0 : K;
        for (int cfg : cfgs) {
            double med[3], best[3];
            std::vector<float> t[3];
            for (int sched = 0; sched < 3; ++sched) {
                if (sched && !((smask >> sched) & 1)) continue;
                g_gemm8_sched = sched;
                _Float16* C = sched == 0 ? C0 : C1;
                CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 2, st));
                if (gemm8_dispatch(cfg, A, Wt, nullptr, nullptr, C, M, N, K, lda, K, N, N, 0, cp, st) != TCL_OK) { printf("launch failed cfg %d\n", cfg); exit(1); }
                CK(hipStreamSynchronize(st));
                hipLaunchKernelGGL(k_ref, dim3(32), dim3(256), 0, st, A, Wt, C, M, N, K, cp, 8192, derr);
                std::vector<float> herr(8192);
                CK(hipMemcpyAsync(herr.data(), derr, 8192 * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
                float mx = 0; for (float v : herr) mx = std::max(mx, v);
                unsigned nd = 0;
                if (sched) {
                    CK(hipMemsetAsync(dcnt, 0, 4, st));
                    hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, st, (const unsigned*)C0, (const unsigned*)C1, (size_t)M * N / 2, dcnt);
                    CK(hipMemcpyAsync(&nd, dcnt, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
                }
                if (mx > 2e-2f || nd) printf("   !!! cfg %d sched %d: max rel err vs f32 reference %.3e, words differing from sched 0: %u\n", cfg, sched, mx, nd);
            }
            for (int round = 0; round < 5; ++round)
                for (int sched = 0; sched < 3; ++sched) {
                    if (!((smask >> sched) & 1)) { t[sched].push_back(1e9f); continue; }
                    g_gemm8_sched = sched;
                    CK(hipEventRecord(e0, st));
                    for (int r = 0; r < 4; ++r) gemm8_dispatch(cfg, A, Wt, nullptr, nullptr, C1, M, N, K, lda, K, N, N, 0, cp, st);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    t[sched].push_back(ms / 4);
                }
            for (int sched = 0; sched < 3; ++sched) { std::sort(t[sched].begin(), t[sched].end()); med[sched] = t[sched][2]; best[sched] = t[sched][0]; }
#ifdef LAB_PROF
            for (int sched = 1; sched < 3; ++sched) {        // segment cycles of the K loop (wave 0 = group 0, wave 4 = group 1), mean over 64 blocks
                g_gemm8_sched = sched;
                unsigned long long* pb; CK(hipMalloc(&pb, 64 * 2 * 5 * 8)); CK(hipMemset(pb, 0, 64 * 2 * 5 * 8));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g8_prof_buf), &pb, sizeof(pb)));
                gemm8_dispatch(cfg, A, Wt, nullptr, nullptr, C1, M, N, K, lda, K, N, N, 0, cp, st); CK(hipStreamSynchronize(st));
                std::vector<unsigned long long> hp(640); CK(hipMemcpy(hp.data(), pb, 640 * 8, hipMemcpyDeviceToHost));
                const int nk = K / 32;
                for (int g = 0; g < 2; ++g) {
                    double sm[5] = {0, 0, 0, 0, 0};
                    for (int b = 0; b < 64; ++b) for (int i = 0; i < 5; ++i) sm[i] += (double)hp[(b * 2 + g) * 5 + i] / 64 / nk;
                    printf("      prof cfg %d sched %d group %d: cycles per K step: barrier1 %.0f | L %.0f | barrier2 %.0f | M %.0f | vmcnt wait %.0f | sum %.0f\n", cfg, sched, g, sm[0], sm[1], sm[2], sm[3], sm[4], sm[0] + sm[1] + sm[2] + sm[3] + sm[4]);
                }
                pb = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g8_prof_buf), &pb, sizeof(pb)));
            }
#endif
            printf("   cfg %d (%s):", cfg, cfg == 1 ? "256x320" : cfg == 2 ? "128x320" : cfg == 3 ? "256x256" : "128x256");
            for (int sched = 0; sched < 3; ++sched) if ((smask >> sched) & 1) printf("  s%d %8.1f us %6.0f TF (best %6.0f)", sched, med[sched] * 1e3, flop / med[sched] / 1e9, flop / best[sched] / 1e9);
            printf("\n");
        }
        // round 4: the 8-phase kernels (gemm8q.hip): bit comparison with the last schedule-0 result in C0, sampled f32 reference, same timing protocol
        for (int qc = 1; qc <= 2; ++qc) {
            if (cfgs.empty() || !gemm8q_ok(qc, M, N, K, lda ? lda : 8, K, N, N, false, 0, cp)) continue;
            CK(hipMemsetAsync(C1, 0xff, (size_t)M * N * 2, st));
            if (gemm8q_dispatch(qc, A, Wt, nullptr, nullptr, C1, M, N, K, lda, K, N, N, 0, cp, st) != TCL_OK) { printf("q launch failed %d\n", qc); exit(1); }
            CK(hipStreamSynchronize(st));
            hipLaunchKernelGGL(k_ref, dim3(32), dim3(256), 0, st, A, Wt, C1, M, N, K, cp, 8192, derr);
            std::vector<float> herr(8192);
            CK(hipMemcpyAsync(herr.data(), derr, 8192 * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            float mx = 0; for (float v : herr) mx = std::max(mx, v);
            unsigned nd = 0;
            CK(hipMemsetAsync(dcnt, 0, 4, st));
            hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, st, (const unsigned*)C0, (const unsigned*)C1, (size_t)M * N / 2, dcnt);
            CK(hipMemcpyAsync(&nd, dcnt, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            if (mx > 2e-2f || nd) printf("   !!! q%d: max rel err vs f32 reference %.3e, words differing from k_gemm8: %u\n", qc, mx, nd);
            std::vector<float> tq;
            for (int round = 0; round < 5; ++round) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < 4; ++r) gemm8q_dispatch(qc, A, Wt, nullptr, nullptr, C1, M, N, K, lda, K, N, N, 0, cp, st);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                tq.push_back(ms / 4);
            }
            std::sort(tq.begin(), tq.end());
            printf("   q%d  (%s, 8-phase):  %8.1f us %6.0f TF (best %6.0f)\n", qc, qc == 1 ? "256x256" : "256x320", tq[2] * 1e3, flop / tq[2] / 1e9, flop / tq[0] / 1e9);
        }
        CK(hipFree(A)); CK(hipFree(Wt)); CK(hipFree(C0)); CK(hipFree(C1));
    }
    return 0;
}
