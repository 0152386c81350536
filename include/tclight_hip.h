/* tclight_hip.h -- C ABI of libtclight_hip.so, the MI355X (gfx950) kernels behind TC-Light's hot paths.
 *
 * The reference (Linketic/TC-Light) is pure Python/PyTorch and has no FFI; its drop-in seams are the
 * Python call signatures listed in SURVEY.md 8(b).  Each entry point below names the reference
 * function (file:line under /root/reference) whose arithmetic it replaces; the tc_light_amd Python modules bind
 * them with ctypes and re-exposes the reference's own signatures (INTEGRATION.md shows the stub).
 *
 * Conventions: all pointers are DEVICE pointers unless the name says host (`sched`, `h_*`);
 * plain C types only; every call takes the HIP stream it enqueues on and returns
 * 0 = TCL_OK, 1 = TCL_EINVAL (bad argument / unsupported shape), 2 = TCL_ELAUNCH (HIP error).
 * Calls are asynchronous w.r.t. the host and never allocate; scratch comes from the caller
 * (`ws`, sized by the matching *_workspace_bytes function).  Re-entrant per stream.
 */
#ifndef TCLIGHT_HIP_H
#define TCLIGHT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================== path 2: two-stage optimiser (f32) */
/* warp_flow(frames, past_flows)  utils/flow_utils.py:5-16 -- bicubic(A=-0.75) backward warp, zeros padding,
 * align_corners.  img/out [n,c,h,w]; flow [n,flow_c>=2,h,w] (first two channels used). */
int tcl_warp_flow_fwd(const float* img, const float* flow, float* out, int n, int c, int h, int w, int flow_c, hipStream_t st);
/* autograd of the above w.r.t. img (grid_sample backward): gimg is zeroed then scatter-added. */
int tcl_warp_flow_bwd(const float* gout, const float* flow, float* gimg, int n, int c, int h, int w, int flow_c, hipStream_t st);
/* clamp(bmm(pixels, M[:3,:3]) + M[:3,3], 0, 1)  generate.py:405-407, utils/dataloader.py:38-42.
 * src [N,3,h,w]; idx int32[nb] frame of each output (NULL = identity); expo [N,3,4]; out [nb,3,h,w]. */
int tcl_apply_exposure(const float* src, const int* idx, const float* expo, float* out, int nb, int h, int w, hipStream_t st);
/* clamp(SH2RGB(features_dc)[unq_inv[frame]], 0, 1)  generate.py:499-501,530-531.  feat [K,3]; inv int32 [N*h*w]. */
int tcl_gather_codebook(const float* feat, const int* inv, const int* fidx, float* out, int nb, int h, int w, hipStream_t st);
/* value[0] = 1 - relaxed_ms_ssim(X, Y, data_range=1, start_level=1)  utils/loss_utils.py:125-211; gradX = d value / dX
 * (NULL to skip).  X, Y: `planes` contiguous h*w planes (= batch*channels). */
size_t tcl_msssim_workspace_bytes(int planes, int h, int w);
int tcl_ms_ssim_loss(const float* X, const float* Y, int planes, int h, int w, float* value, float* gradX, void* ws, hipStream_t st);
/* TVLoss(weight)(x) and its gradient  utils/loss_utils.py:324-340.  ws16: 16 bytes of scratch. */
int tcl_tv_loss(const float* x, int b, int c, int h, int w, float weight, float* value, float* grad, void* ws16, hipStream_t st);
/* torch.optim.Adam single-tensor step (generate.py:381,483-487); g is consumed and zeroed. step counts from 1. */
int tcl_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step, hipStream_t st);
/* RGB2SH(torch_scatter.scatter(pixels, unq_inv, reduce='mean'))  generate.py:477-479.  cnt: K floats scratch. */
int tcl_scatter_mean_rgb2sh(const float* img, const int* inv, float* feat, float* cnt, int n, int h, int w, size_t K, hipStream_t st);

/* Whole-stage drivers: every iteration is enqueued on `st`; no host synchronisation inside.
 * sched (HOST) int32 [iters][batch]: frame ids of each mini-batch, -1 pads a short batch (stands in for
 * DataLoader(shuffle=True), generate.py:363-367).  d_cat (DEVICE) int32 [iters][2*batch]: per iteration
 * [cur(b) | max(cur-1,0)(b) | pad].  losses: device float [iters]. */
size_t tcl_stage_workspace_bytes(int batch, int h, int w);
/* Generator.exposure_align  generate.py:354-451.  exposure [N,3,4] (= eye on entry), g/m/v zero on entry;
 * aligned_out [N,3,h,w] receives OptDataset.exposure_align's result.  lr(it) = get_expon_lr_func(lr_init, lr_final,
 * max_steps = epochs*N/batch)((it / iters_per_epoch) * N / batch + it % iters_per_epoch + 1)  (generate.py:372,394). */
int tcl_exposure_align(const float* edited, const float* flows, const float* masks, int N, int H, int W, const int* sched,
                       const int* d_cat, int iters, int iters_per_epoch, int batch, int epochs, float lr_init, float lr_final, float lambda_dssim,
                       float lambda_flow, float* exposure, float* g, float* m, float* v, float* losses, float* aligned_out,
                       void* ws, hipStream_t st);
/* Generator.unique_tensor_optimization  generate.py:453-533.  feat [K,3] initialised by tcl_scatter_mean_rgb2sh;
 * images_out [N,3,h,w] (may be NULL) receives the final gather. */
int tcl_unique_tensor_opt(const float* target, const float* flows, const float* masks, const int* unq_inv, int N, int H, int W,
                          size_t K, const int* sched, const int* d_cat, int iters, int batch, float feature_lr, float lambda_dssim,
                          float lambda_flow, float lambda_tv, float* feat, float* g, float* m, float* v, float* losses,
                          float* images_out, void* ws, hipStream_t st);

#ifdef __cplusplus
}
#endif
#endif
