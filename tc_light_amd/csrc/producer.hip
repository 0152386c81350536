// Stage-2 input producer on gfx950: soft forward/backward-consistency masks and flow (track) ids.
//   get_soft_mask_bwds  utils/flow_utils.py:40-54     get_flowid  utils/flow_utils.py:56-93
//   voxelization(voxel_size=None)  utils/general_utils.py:222-256  (== the ids themselves: they are already dense)
// get_flowid is sequential over frames; each frame is four launches (clear, splat with a deterministic 64-bit atomicMax, a single-pass
// decoupled-look-back exclusive scan of the "unassigned" flags fused with the id assignment, counter bump) with the running id counter
// kept on the device -- no host sync, no library primitive.  Write conflicts (several source
// pixels landing on one target) are "last writer wins" in the reference (nondeterministic on its GPU path); here the largest
// source index wins, which is what its CPU path yields.
#include "common.h"
#include "bicubic.h"
#include "../../include/tclight_hip.h"

__global__ void k_soft_mask(const float* __restrict__ img, const float* __restrict__ fwd, const float* __restrict__ past, int H, int W,
                            float alpha, float beta, float thr_abs, float* __restrict__ mask) {
    const int P = H * W, n = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        float m = 1.f;
        if (n > 0) {
            const int y = p / W, x = p - y * W;
            const float* pf = past + (size_t)n * 2 * P;
            const float px = pf[p], py = pf[P + p];
            Tap t = make_tap(px, py, x, y, W, H);
            const float* f0 = fwd + (size_t)(n - 1) * 2 * P; const float* i0 = img + (size_t)(n - 1) * 3 * P;
            float fx = 0, fy = 0, c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int yy = t.y0 + j; if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int xx = t.x0 + i; if (xx < 0 || xx >= W) continue;
                    float w = t.wy[j] * t.wx[i]; int a = yy * W + xx;
                    fx += w * f0[a]; fy += w * f0[P + a]; c0 += w * i0[a]; c1 += w * i0[P + a]; c2 += w * i0[2 * P + a];
                }
            }
            float n1 = sqrtf((px + fx) * (px + fx) + (py + fy) * (py + fy));
            float n2 = (sqrtf(px * px + py * py) + sqrtf(fx * fx + fy * fy) + 1.f) * alpha;
            m = 1.f / (1.f + __expf(beta * (n1 - n2)));
            const float* i1 = img + (size_t)n * 3 * P;
            float d = fmaxf(fmaxf(fabsf(c0 - i1[p]), fabsf(c1 - i1[P + p])), fabsf(c2 - i1[2 * P + p]));
            m *= 1.f / (1.f + __expf(beta * (d - thr_abs)));
        }
        mask[(size_t)n * P + p] = m;
    }
}

__global__ void k_flowid_init(int* __restrict__ ids, int P, int* __restrict__ last_id) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) ids[p] = p;
    if (blockIdx.x == 0 && threadIdx.x == 0) *last_id = P;
}
// forward-splat the ids of frame i-1 to round(grid + flow_{i-1}) (flow_utils.py:78-88)
__global__ void k_flowid_splat(const int* __restrict__ ids_prev, const float* __restrict__ f_prev, const float* __restrict__ f_cur,
                               const float* __restrict__ flow, const float* __restrict__ mask_cur, int H, int W, float thr,
                               unsigned long long* __restrict__ keys) {
    const int P = H * W;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int y0 = p / W, x0 = p - y0 * W;
        const int x = (int)rintf((float)x0 + flow[p]), y = (int)rintf((float)y0 + flow[P + p]);     // torch.round: half to even
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        if (!(mask_cur[p] > 0.5f)) continue;                  // NB: the mask is tested at SOURCE coordinates (flow_utils.py:83)
        const int q = y * W + x;
        float d = fmaxf(fmaxf(fabsf(f_cur[q] - f_prev[p]), fabsf(f_cur[P + q] - f_prev[P + p])), fabsf(f_cur[2 * P + q] - f_prev[2 * P + p]));
        if (!(d < thr)) continue;
        atomicMax(keys + q, ((unsigned long long)(p + 1) << 32) | (unsigned)ids_prev[p]);
    }
}
// Exclusive scan of the flags (keys[p] == 0) fused with the assignment ids[p] = key ? carried id : *last_id + (#unassigned before p).
// Single pass, decoupled look-back: tiles of 1024 pixels take their index from an atomic ticket (so every lower tile is already resident),
// publish (state << 32 | value) in one 64-bit word -- state 1 = the tile's own count, 2 = inclusive prefix -- and a tile sums its
// predecessors' words backwards until it meets a prefix.  One relaxed agent-scope atomic per word: the value travels inside it.
#define SCAN_TILE 1024
__global__ __launch_bounds__(256) void k_flowid_scan_assign(const unsigned long long* __restrict__ keys, int* __restrict__ ids, int P,
                                                            const int* __restrict__ last_id, unsigned long long* __restrict__ status,
                                                            unsigned* __restrict__ ticket) {
    __shared__ int s_tile, s_prefix, s_wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile, p0 = tile * SCAN_TILE + tid * 4;
    unsigned long long kk[4]; int f[4], loc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { kk[j] = p0 + j < P ? keys[p0 + j] : 1ull; f[j] = kk[j] == 0ull; loc += f[j]; }
    int incl = loc;                                   // inclusive scan of the per-thread counts over the wave, then over the 4 waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int woff = 0;
    for (int j = 0; j < wv; ++j) woff += s_wsum[j];
    const int total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    if (tid == 0) {
        int run = 0;
        if (tile > 0) {
            __hip_atomic_store(status + tile, (1ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int j = tile - 1;; --j) {
                unsigned long long w;
                do { w = __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> 32) == 0ull);
                run += (int)(unsigned)w;
                if ((w >> 32) == 2ull) break;
            }
        }
        __hip_atomic_store(status + tile, (2ull << 32) | (unsigned)(run + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = run;
    }
    __syncthreads();
    int off = *last_id + s_prefix + woff + incl - loc;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (p0 + j < P) { ids[p0 + j] = kk[j] ? (int)(kk[j] & 0xFFFFFFFFull) : off; off += f[j]; }
}
__global__ void k_flowid_bump(const unsigned long long* __restrict__ status, int ntiles, int* __restrict__ last_id) {
    *last_id += (int)(unsigned)status[ntiles - 1];       // inclusive prefix of the last tile = fresh ids of this frame
}

extern "C" {

int tcl_soft_mask_bwds(const float* img, const float* fwd, const float* past, int N, int H, int W, float alpha, float beta, float thr_abs,
                       float* mask, hipStream_t st) {
    TCL_CHECK_ARG(img && fwd && past && mask && N > 0 && H > 1 && W > 1);
    int g = stream_grid((long)H * W, 256, 2); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_soft_mask, dim3(g, N), dim3(256), 0, st, img, fwd, past, H, W, alpha, beta, thr_abs, mask);
    TCL_LAUNCH_RET();
}
size_t tcl_flowid_workspace_bytes(int H, int W) {
    const size_t P = (size_t)H * W, nt = (P + SCAN_TILE - 1) / SCAN_TILE;
    return P * 8 + (nt + 2) * 8 + 1024;                // keys | scan status words | ticket
}
int tcl_flowid(const float* frames, const float* fwd_flows, const float* masks, int N, int H, int W, float thr_abs, int* ids, int* last_id,
               void* ws, hipStream_t st) {
    TCL_CHECK_ARG(frames && fwd_flows && masks && ids && last_id && ws && N > 0 && (size_t)N * H * W < 0x7FFFFFFFull);
    const int P = H * W, nt = (P + SCAN_TILE - 1) / SCAN_TILE;
    unsigned long long* keys = (unsigned long long*)ws;
    unsigned long long* status = keys + P;
    unsigned* ticket = (unsigned*)(status + nt);
    int g = stream_grid(P, 256, 2); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_flowid_init, dim3(g), dim3(256), 0, st, ids, P, last_id);
    for (int i = 1; i < N; ++i) {
        if (hipMemsetAsync(keys, 0, (size_t)P * 8 + (size_t)(nt + 1) * 8, st) != hipSuccess) return TCL_ELAUNCH;      // keys, status, ticket
        hipLaunchKernelGGL(k_flowid_splat, dim3(g), dim3(256), 0, st, ids + (size_t)(i - 1) * P, frames + (size_t)(i - 1) * 3 * P, frames + (size_t)i * 3 * P,
                           fwd_flows + (size_t)(i - 1) * 2 * P, masks + (size_t)i * P, H, W, thr_abs, keys);
        hipLaunchKernelGGL(k_flowid_scan_assign, dim3(nt), dim3(256), 0, st, keys, ids + (size_t)i * P, P, last_id, status, ticket);
        hipLaunchKernelGGL(k_flowid_bump, dim3(1), dim3(1), 0, st, status, nt, last_id);
    }
    TCL_LAUNCH_RET();
}

}  // extern "C"
