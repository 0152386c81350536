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
/* clamp(SH2RGB(features_dc)[unq_inv[frame]], 0, 1)  generate.py:499-501,530-531.  The codebook is stored CHANNEL-PLANAR,
 * feat [3,K] (= features_dc.t()), so gathers / scatter-adds of neighbouring pixels are contiguous per channel; inv int32 [N*h*w]. */
int tcl_gather_codebook(const float* feat, const int* inv, const int* fidx, float* out, int nb, int h, int w, size_t K, hipStream_t st);
/* value[0] = 1 - relaxed_ms_ssim(X, Y, data_range=1, start_level=1)  utils/loss_utils.py:125-211; gradX = d value / dX
 * (NULL to skip).  X, Y: `planes` contiguous h*w planes (= batch*channels). */
size_t tcl_msssim_workspace_bytes(int planes, int h, int w);
int tcl_ms_ssim_loss(const float* X, const float* Y, int planes, int h, int w, float* value, float* gradX, void* ws, hipStream_t st);
/* TVLoss(weight)(x) and its gradient  utils/loss_utils.py:324-340.  ws16: 2 KiB of scratch (fixed-point loss accumulators). */
int tcl_tv_loss(const float* x, int b, int c, int h, int w, float weight, float* value, float* grad, void* ws16, hipStream_t st);
/* torch.optim.Adam single-tensor step (generate.py:381,483-487); g is consumed and zeroed. step counts from 1. */
int tcl_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step, hipStream_t st);
/* How the flow term scatters d(loss)/d(warped previous frame) (generate.py:420-427 / :507-514 backward): 1 (default) = through an LDS window per
 * 64x16 pixel tile, flushed once; 0 = every addend straight to global memory.  Integer (fixed-point) sums either way: same bits when W % 64 == 0,
 * the same value up to the f32 rounding of differently grouped partial sums otherwise.  Test / A-B hook (env TCL_FLOW_TILED sets the initial mode). */
int tcl_flow_scatter_mode(int tiled);
/* Per-frame fixed-point scale of that scatter (the reference's float grid_sample backward, utils/flow_utils.py:5-16, has no range limit; the
 * engine's bit-reproducible integer cells do).  flow_shift[f] = min(22, 30 - ceil(log2 L_f)) with L_f = the largest number of masked-in pixels
 * of frame f whose bicubic 4x4 window covers one pixel of frame f-1: with |mask * weight| <= 1 no 32-bit cell can overflow, however strongly the
 * flow converges.  Call once per clip (flows / masks are constants of both stages) and pass flow_shift to the four stage entry points below.
 * flows [N,2,h,w], masks [N,1,h,w]; scratch: (h*w + 1) int32; flow_shift: N int32 (device). */
int tcl_flow_cell_shift(const float* flows, const float* masks, int N, int H, int W, int* scratch, int* flow_shift, hipStream_t st);
/* Are the track ids of every frame pairwise distinct (true of get_flowid's output, utils/flow_utils.py:56-93: a pixel of frame i takes the id
 * of ONE pixel of frame i-1 or a fresh id)?  *result (device int) <- 1 / 0; scratch: K ints.  Stage 2 then accumulates codebook rows one frame
 * at a time without atomics -- bit-reproducible -- by passing ids_unique = 1 below; with 0 it falls back to float atomics (any id layout). */
int tcl_track_ids_unique(const int* unq_inv, int N, int H, int W, size_t K, int* scratch, int* result, hipStream_t st);
/* RGB2SH(torch_scatter.scatter(pixels, unq_inv, reduce='mean'))  generate.py:477-479 -> feat [3,K] planar.  cnt: K floats scratch. */
int tcl_scatter_mean_rgb2sh(const float* img, const int* inv, float* feat, float* cnt, int n, int h, int w, size_t K, int ids_unique, hipStream_t st);

/* Whole-stage drivers: every iteration is enqueued on `st`; no host synchronisation inside.
 * sched (HOST) int32 [iters][batch]: frame ids of each mini-batch, -1 pads a short batch (stands in for
 * DataLoader(shuffle=True), generate.py:363-367).  d_cat (DEVICE) int32 [iters][2*batch]: per iteration
 * [cur(b) | max(cur-1,0)(b) | pad].  losses: device float [iters]. */
size_t tcl_stage_workspace_bytes(int batch, int h, int w);
/* Generator.exposure_align  generate.py:354-451.  exposure [N,3,4] (= eye on entry), g/m/v zero on entry;
 * aligned_out [N,3,h,w] receives OptDataset.exposure_align's result.  lr(it) = get_expon_lr_func(lr_init, lr_final,
 * max_steps = epochs*N/batch)((it / iters_per_epoch) * N / batch + it % iters_per_epoch + 1)  (generate.py:372,394). */
int tcl_exposure_align(const float* edited, const float* flows, const float* masks, const int* flow_shift, int N, int H, int W, const int* sched,
                       const int* d_cat, int iters, int iters_per_epoch, int batch, int epochs, float lr_init, float lr_final, float lambda_dssim,
                       float lambda_flow, float* exposure, float* g, float* m, float* v, float* losses, float* aligned_out,
                       void* ws, hipStream_t st);
/* Generator.unique_tensor_optimization  generate.py:453-533.  feat/g/m/v [3,K] planar, feat initialised by tcl_scatter_mean_rgb2sh;
 * images_out [N,3,h,w] (may be NULL) receives the final gather. */
int tcl_unique_tensor_opt(const float* target, const float* flows, const float* masks, const int* flow_shift, const int* unq_inv, int N, int H, int W,
                          size_t K, int ids_unique, const int* sched, const int* d_cat, int iters, int batch, float feature_lr, float lambda_dssim,
                          float lambda_flow, float lambda_tv, float* feat, float* g, float* m, float* v, float* losses,
                          float* images_out, void* ws, void* lazy_ws, hipStream_t st);
/* lazy_ws (may be NULL): tcl_stage2_lazy_workspace_bytes(K, iters) bytes.  With it (and ids_unique) the dense Adam of generate.py:483-487 is
 * applied LAZILY: a codebook row is only read / written in the iterations whose mini-batch holds it, the steps it skipped in between (pure
 * momentum decay, no gradient) are replayed with the same arithmetic right before it is needed, and one dense pass ends the stage -- the
 * result is bit-identical to stepping all K rows every iteration, at ~(rows of the mini-batch) / K of the optimiser's HBM traffic. */
size_t tcl_stage2_lazy_workspace_bytes(size_t K, int iters);

/* One mini-batch of stage 1 / stage 2, GRADIENT ONLY (forward + backward of generate.py:396-429 / :496-522, no optimiser step): what the
 * whole-stage drivers above run per iteration, exposed so that a multi-GPU host loop can put a collective between the gradient and the
 * Adam step (tc_light_amd/post_opt.py; SURVEY 8(e): the slots of a mini-batch are dealt to the ranks, codebook / exposure gradients meet
 * in reduce_scatter / all_reduce over xGMI, tcl_adam_step updates).  d_cidx (DEVICE) int32 [2*b_loc] = [cur(b_loc) | max(cur-1,0)(b_loc)]
 * for THIS caller's slots; b_glob = slots of the whole mini-batch, nvalid_glob = how many of them have frame id > 0: every mean of the
 * reference's loss runs over the global mini-batch, so the callers' partial gradients and *loss_part values simply add up.
 * g is accumulated into (+=) and must be zero (or hold other slots' partial sums) on entry; ws: tcl_stage_workspace_bytes(b_loc, h, w). */
int tcl_exposure_grad(const float* edited, const float* flows, const float* masks, const int* flow_shift, int N, int H, int W, const int* d_cidx, int b_loc,
                      int b_glob, int nvalid_glob, float lambda_dssim, float lambda_flow, const float* exposure, float* g,
                      float* loss_part, void* ws, hipStream_t st);
int tcl_unique_tensor_grad(const float* target, const float* flows, const float* masks, const int* flow_shift, const int* unq_inv, int N, int H, int W, size_t K,
                           int ids_unique, const int* d_cidx, int b_loc, int b_glob, int nvalid_glob, float lambda_dssim, float lambda_flow,
                           float lambda_tv, const float* feat, float* g, float* loss_part, void* ws, hipStream_t st);

/* ===================================================================== path 1: denoising loop (f16, f32 accumulate)
 * Activations are NHWC / token-major [B, H*W, C] f16 on the device.  These replace the library kernels the reference
 * reaches through diffusers' UNet2DConditionModel / AutoencoderKL (generate.py:342-347; generate_utils.py:144,161). */
/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + resid[M,N];  act: 0 none, 1 SiLU.  K % 64 == 0; lda, ldw (row strides of A, W
 * in halves) % 8 == 0.  torch.nn.Linear / 1x1 Conv2d.  bias / resid may be NULL.
 * act: 0 none, 1 SiLU, 3 ReLU, 4 GELU (erf), applied to A.W^T + bias before the residual is added; 5 = GELU applied AFTER the residual
 * add (C = gelu(A.W^T + bias + resid), resid required).
 * act 2 = fused GEGLU (diffusers ff.net.0 + GEGLU): W/bias rows must be pre-arranged in 64-row groups [32 value rows | the 32
 * matching gate rows]; C is [M, N/2] = value * gelu(gate); N % 64 == 0, no resid. */
int tcl_gemm_f16(const void* A, const void* W, const void* bias, const void* resid, void* C, int M, int N, int K, int lda, int ldw,
                 int ldc, int ldr, int act, hipStream_t st);
/* Register caller-owned device scratch for split-K partial sums (used by tcl_gemm_f16 / tcl_conv3x3_f16 when the tile grid
 * would leave most CUs idle).  NULL disables split-K.  All calls that use it must be issued on one stream. */
int tcl_set_workspace(void* ws, size_t bytes);
/* Tuning / test hook: force the kernel configuration (cfg ids in csrc/gemm.hip; 0 = automatic choice) and the number of K
 * splits (0/1 = none) for every following tcl_gemm_f16 / tcl_conv3x3_f16 call; calls the forced configuration cannot serve
 * return TCL_EINVAL.  Process-global, not thread-safe. */
int tcl_gemm_tune(int cfg, int splits);
/* Automatic configuration choice (default on): the first call with a new problem shape times the candidate tile
 * configurations on the caller's stream (synchronising it once) and caches the winner for the life of the process; results are
 * bit-identical whichever candidate wins (the K-split count is a fixed function of the shape).  0 = static heuristic only and
 * drops the cache. */
int tcl_gemm_autotune(int enable);
/* Persistent tile table (text, one line per problem shape).  _load merges a file into the cache (entries are re-validated against each
 * call's leading dimensions before use); _save writes the cache; _size = entries.  tcl_gemm_autotune(2) = table-only mode: shapes missing
 * from the table take the static heuristic instead of being timed -- no host synchronisation in the hot loop and the same tile (hence the
 * same bits, run to run) for a given shape.  tc_light_amd/unet.py loads tc_light_amd/gemm_tune_gfx950.txt at start-up when it exists. */
int tcl_gemm_tune_save(const char* path);
int tcl_gemm_tune_load(const char* path);
size_t tcl_gemm_tune_size(void);
/* 3x3 Conv2d as implicit GEMM on NHWC: X [B,Hin,Win,Cin], W [Cout, 9*Cin] (tap-major: (ky*3+kx)*Cin + c), Y [B,Hout,Wout,Cout].
 * pad=1: padding 1 (UNet ResnetBlock2D / Downsample2D stride 2); pad=0 with stride 2: the VAE encoder's (0,1,0,1) padding.
 * Hup/Wup > 0: the input is first nearest-upsampled to Hup x Wup (Upsample2D with explicit output size), fused in the gather. */
int tcl_conv3x3_f16(const void* X, const void* W, const void* bias, const void* resid, void* Y, int B, int Hin, int Win, int Cin,
                    int Cout, int stride, int pad, int Hup, int Wup, int act, hipStream_t st);

/* torch.nn.GroupNorm(groups, C1+C2, eps) [+ SiLU] over x = cat([x1, x2], channel) (x2 may be NULL with C2 = 0): the
 * ResnetBlock2D / Transformer2DModel / AutoencoderKL norms, with the up-block skip concat folded in.  y [B,HW,C1+C2].
 * ws: tcl_groupnorm_workspace_bytes(B, C) bytes of scratch (per-block partial sums).  Deterministic: no float atomics -- the same
 * input gives the same bits on every run. */
size_t tcl_groupnorm_workspace_bytes(int B, int C);
int tcl_groupnorm_f16(const void* x1, int C1, const void* x2, int C2, const void* gamma, const void* beta, void* y, int B, int HW,
                      int groups, float eps, int silu, void* ws, hipStream_t st);
/* The same with a second output: yraw [B,HW,C1+C2] (may be NULL) receives the un-normalised channel concat cat([x1, x2]) -- the operand of the
 * up-block ResnetBlock2D's conv_shortcut (diffusers ResnetBlock2D.forward: `input_tensor = conv_shortcut(input_tensor)` on the concatenated
 * input) -- from the pass that reads x1 / x2 anyway, instead of a concat pass of its own (tcl_concat_channels_f16). */
int tcl_groupnorm_concat_f16(const void* x1, int C1, const void* x2, int C2, const void* gamma, const void* beta, void* y, void* yraw, int B, int HW,
                             int groups, float eps, int silu, void* ws, hipStream_t st);
/* torch.nn.LayerNorm(C) (BasicTransformerBlock.norm1/2/3). */
int tcl_layernorm_f16(const void* x, const void* gamma, const void* beta, void* y, long rows, int C, float eps, hipStream_t st);
/* The same, plus metric = y / |y| per row with tcl_tome_normalize_f16's f16 arithmetic (norm1 of a VidToMe-patched block, patch.py:161-166 ->
 * merge.py:84: the matching metric of the block's tokens, produced while they are in registers). */
/* C = act(LayerNorm(x; gamma, beta, eps) . W^T + bias) (+ resid): norm2 -> attn2.to_q and norm3 -> ff.net.0 (GEGLU, act 2) of BasicTransformerBlock
 * (generate.py:342-347 -> diffusers) in one kernel, for K = 320 (the level-0 blocks; other widths: tcl_layernorm_f16 + tcl_gemm_f16).  The
 * normalised activations are rounded to f16 exactly as tcl_layernorm_f16 rounds them; the f32 statistics are summed in another order (not bit-identical
 * to the two-kernel route, 1e-3 rel-L2 like any two f16 LayerNorms).  N >= 128, N % 32 == 0 (GEGLU: % 64), 16-B aligned rows. */
int tcl_ln_gemm_f16(const void* x, const void* gamma, const void* beta, float eps, const void* W, const void* bias, const void* resid, void* C,
                    int M, int N, int K, int ldx, int ldw, int ldc, int ldr, int act, hipStream_t st);
/* The attn2 `to_q` projection of a C = 320 transformer block written straight into the attention kernel's query panel (diffusers Attention.to_q ->
 * AttnProcessor2_0, utils/model_utils.py:66-67): LayerNorm(x) @ W^T -> ws_q's Qp [B, H, ceil256(Tq), 48], pre-scaled by scale * log2 e, exactly as
 * tcl_attention_pack_f16 would have packed the Linear's output (same rounding points), without the [M, H d] round trip.  M = B * Tq rows, d = 40.
 * Rows Tq .. ceil256(Tq) of every (b, h) panel are NOT written: the caller keeps ws_q (tcl_attention_q_bytes) zero-initialised and reuses it.
 * Follow with tcl_attention_f16(..., pack_kv = 4 (pre-packed), ws_q, ws_kv). */
int tcl_ln_gemm_qpanel_f16(const void* x, const void* gamma, const void* beta, float eps, const void* W, int M, int H, int d, int Tq, int ldx, int ldw,
                           float scale, void* ws_q, hipStream_t st);
/* attn1's fused QKV projection of a VidToMe-merging block written straight into the attention panels (patch.py:170-176: `attn1(norm_hidden_states)` over the
 * merged tokens; diffusers Attention.to_q / to_k / to_v, no bias): x [ne*T, K] @ W[3 H d, K]^T -> Qp / Kp / V^T panels in ws_q / ws_kv with exactly the
 * (entry b's rows start x_bs elements after entry b-1's; row_index != NULL: token t is row row_index[t] of its entry's block -- the merge map of
 * merge.py's "replace" mode applied in the operand load instead of by a gather pass)
 * bytes tcl_attention_pack_f16 would have written from the [ne*T, 3 H d] product (d = 40 or 80).  ws_q / ws_kv: tcl_attention_q_bytes / _kv_bytes(ne, H, T, d),
 * ZERO-INITIALISED once per (ne, H, T, d) by the caller and reusable by stream-ordered calls of that shape (padding is never written).  Follow with
 * tcl_attention_f16(..., pack_kv = 4 (pre-packed) [| 2], ws_q, ws_kv). */
int tcl_gemm_qkv_panels_f16(const void* x, long x_bs, const int* row_index, const void* W, int ne, int T, int H, int d, int K, int ldx, int ldw, float scale,
                            void* ws_q, void* ws_kv, hipStream_t st);
int tcl_layernorm_metric_f16(const void* x, const void* gamma, const void* beta, void* y, void* metric, long rows, int C, float eps, hipStream_t st);
/* diffusers GEGLU: in [rows, 2D] -> out [rows, D] = in[:, :D] * gelu(in[:, D:]) (exact erf gelu). */
int tcl_geglu_f16(const void* in, void* out, long rows, int D, hipStream_t st);
/* in-place softmax(scale * x) over rows of length T (VAE mid-block single-head attention). */
int tcl_softmax_rows_f16(void* x, long rows, int T, int ld, float scale, hipStream_t st);
int tcl_concat_channels_f16(const void* x1, int C1, const void* x2, int C2, void* y, long rows, hipStream_t st);
/* im2col for 3x3/pad-1 convs with tiny Cin (IC-Light conv_in 8->320, utils/model_utils.py:21-26; VAE 3->128, 4->512). */
int tcl_im2col3x3_small_f16(const void* x, void* out, int B, int H, int W, int Cin, int Kpad, hipStream_t st);
/* y = f16(act_out(W . act_in(x) + bias)) + add : time_embedding MLP and ResnetBlock2D.time_emb_proj (M = 1). */
int tcl_gemv_f16(const void* W, const void* x, const void* bias, const void* add, void* y, int N, int K, int silu_in, int silu_out, hipStream_t st);
/* diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]. */
int tcl_timestep_embed_f16(float t, int dim, void* out, hipStream_t st);
/* Generator.pred_noise input assembly (generate.py:295-298 + model_utils.py:35-40): latents x / concat_conds [N,4,h,w] f16 ->
 * UNet input [2F,H',W',8] NHWC.  mode 0: xy-plane, idx = frame ids; mode 1: yt-plane ('n c h w -> w c n h', generate.py:267),
 * idx = latent columns, frames sl..sl+nwin. */
int tcl_pack_latents_f16(const void* x, const void* cond, const int* idx, int F, int mode, int sl, int nwin, int h, int w, void* out, hipStream_t st);
/* CFG combine uncond + g*(cond - uncond) (generate.py:349-350) scattered back to noises[N,4,h,w]; in mode 1 frames
 * n < scale_upto are multiplied by `scale` (the sqrt(0.5) overlap rule, generate.py:276-278) and only the first nkeep
 * frames of the window are written (the rest is overwritten by the next window in the reference's sequential order). */
int tcl_unpack_cfg_f16(const void* eps, const int* idx, int F, int mode, int sl, int nwin, int h, int w, float guidance, int scale_upto,
                       float scale, int nkeep, void* noise, hipStream_t st);
/* noises_t <- AdaIN(noises_t, noises); noises <- sqrt(a)*noises_t + sqrt(1-a)*noises  (generate.py:281-282,
 * utils/general_utils.py:137-156).  planes = N*4, hw = h*w. */
int tcl_adain_fuse_f16(void* noises_t, void* noises, int planes, int hw, float alpha, hipStream_t st);
/* DPMSolverMultistepScheduler.step, sde-dpmsolver++ (generate.py:235): m0 <- x0 = (x - sigma_t*eps)/alpha_t;
 * x <- ca*x + cb0*m0 + cb1*m1 + cc*z (coefficients from tc_light_amd/scheduler.py; f32 math; m1/z may be NULL). */
int tcl_dpm_sde_step_f16(void* x, const void* eps, float* m0, const float* m1, const void* z, long n, float sigma_t, float alpha_t, float ca,
                         float cb0, float cb1, float cc, hipStream_t st);
/* layout conversions around the VAE (generate_utils.py:140-172): 2*img-1 -> NHWC8 f16; clamp(y/2+0.5) -> [B,3,H,W] f32. */
int tcl_img_to_nhwc8_f16(const float* img, void* out, int B, int HW, hipStream_t st);
int tcl_nhwc_to_img_f32(const void* y, int ldc, float* img, int B, int HW, hipStream_t st);
int tcl_nhwc_to_nchw_f16(const void* y, int ldc, void* out, int B, int C, int HW, float scale, hipStream_t st);
int tcl_nchw_to_nhwc_f16(const void* x, void* out, int ldc, int B, int C, int HW, float scale, hipStream_t st);
int tcl_transpose_f16(const void* in, void* out, int batch, int R, int Cc, int ldi, int ldo, hipStream_t st);
/* 1x1 Conv2d with <= 8 channels (AutoencoderKL quant_conv 8->8 / post_quant_conv 4->4): y[m, :Co] = W[Co,Ci] x[m, :Ci] + b;
 * columns Co..ldo-1 of y are zeroed. */
int tcl_conv1x1_small_f16(const void* x, int ldi, const void* W, const void* b, void* y, int ldo, long M, int Ci, int Co, hipStream_t st);

/* softmax(Q K^T * scale) V per head, flash style (torch SDPA / xformers via AttnProcessor2_0: attn1 on the VidToMe-merged
 * tokens, patch.py:170-176, and attn2 text cross-attention).  q/k/v/o point at head 0 of batch 0 with heads interleaved
 * in channels (head hh = channels [hh*d, (hh+1)*d)); ld* row strides and *bs batch strides in halves; d in {40, 80, 160}.
 * K/V batch = b / kv_div.  pack_kv bit 0: 0 reuses the K/V panels a previous call left in ws_kv (text K/V are constant per run); bit 1: the
 * B samples are one half of an identical pair (the CFG halves before the first text cross-attention): the kernel variant is chosen as for 2 B
 * samples, so the half alone gives the bits the full batch would have given; bit 2: the panels in ws_q / ws_kv were already written by
 * tcl_attention_pack_f16 (same arguments; e.g. on another stream, with the caller's event between the two): only the attention kernels run.
 * ws_q (tcl_attention_q_bytes): packed Q panel + one int per 128-query block (head_dim 40, large launches: blocks whose speculative
 * softmax left the f16 range are flagged there and redone by the exact-maximum kernel of the same call); no initialisation needed. */
size_t tcl_attention_q_bytes(int B, int H, int Tq, int d);
size_t tcl_attention_kv_bytes(int Bkv, int H, int Tk, int d);
int tcl_attention_f16(const void* q, int ldq, long qbs, const void* k, int ldk, long kbs, const void* v, int ldv, long vbs, void* o, int ldo,
                      long obs, int B, int H, int Tq, int Tk, int d, float scale, int kv_div, int pack_kv, void* ws_q, void* ws_kv,
                      hipStream_t st);
/* The packing half of tcl_attention_f16 alone: Q panel (scaled) into ws_q and, with pack_kv bit 0, the K / V^T panels into ws_kv. */
int tcl_attention_pack_f16(const void* q, int ldq, long qbs, const void* k, int ldk, long kbs, const void* v, int ldv, long vbs, int B, int H, int Tq,
                           int Tk, int d, float scale, int kv_div, int pack_kv, void* ws_q, void* ws_kv, hipStream_t st);

/* ---- VidToMe token merging (utils/VidToMe/vidtome/merge.py, patch.py:14-91); int32 maps live on the device ---- */
/* metric / metric.norm(dim=-1) with f16 rounding of the norm and the quotient (merge.py:84, :386). */
int tcl_tome_normalize_f16(const void* x, void* y, long rows, int C, hipStream_t st);
/* bipartite soft matching with align_batch (merge.py:84-117 randframe, :389-421 2s): cosine scores of src rows a_pos[na]
 * vs dst rows b_pos[nb] of metric [Bt,T,C] (normalised), batch entries concatenated along dst, greedy row max; the r src rows with
 * the highest f16 score are merged (ties: highest score, lowest concatenated dst index for the match, lowest src index for the cut).
 * The score matrix is never materialised and nothing is sorted: unmerged src rows keep their index order in the merged sequence (the
 * reference orders them by score, which only permutes the tokens of a permutation-invariant attention).
 * Outputs: mrg[na-r+nb] = input position feeding each merged slot ([unmerged src | dst], mode "replace");
 *          unm[position] = merged slot each input position is restored from (merge.py:135-155).
 * ws: tcl_tome_match_workspace_bytes(na) bytes, ZEROED ONCE by the caller before the first call and then only passed to this function
 * (one stream; may be re-used for any smaller na): every call leaves it all-zero again except two result words at a fixed offset (layout:
 * 4 KiB control | 65 536 score-histogram bins | keys).  Per call: the score kernel + two small launches (threshold select, maps). */
size_t tcl_tome_match_workspace_bytes(int na);
/* 1: C = 640 affine matches take the strip-resident kernel too (same maps; faster alone, slower beside the flash kernel: off by default; TCL_TOME640). */
int tcl_tome_strip640(int enable);
int tcl_tome_match_f16(const void* metric, long bstride, int Bt, int C, const int* a_pos, int na, const int* b_pos, int nb, int r,
                       int* mrg, int* unm, void* ws, hipStream_t st);
/* The same with a hint: the positions are affine -- a_pos[i] = i < a_split ? i : i + a_gap and b_pos[j] = b0 + j -- which is what every VidToMe
 * match is (the dst frame of a random-frame merge, the two halves of a two-set merge).  For C = 320 this takes a kernel that keeps a src
 * strip in registers and sweeps the dst tiles with a running maximum (csrc/merge.hip, k_tome_match320); same outputs, bit for bit. */
int tcl_tome_match_affine_f16(const void* metric, long bstride, int Bt, int C, const int* a_pos, int na, const int* b_pos, int nb, int r,
                              int a_split, int a_gap, int b0, int* mrg, int* unm, void* ws, hipStream_t st);
/* out[i] = outer[off + inner[i]] (inner NULL = identity): composition of unmerge maps (func_warper, vidtome/utils.py:42-48). */
int tcl_index_compose(const int* outer, const int* inner, int off, int n, int* out, hipStream_t st);
/* out[b][p] = map[p] >= 0 ? s1[b][map[p]] : s2[b][~map[p]]  (merge in "replace" mode; map NULL = copy). */
int tcl_gather_rows_f16(const void* s1, long bs1, const void* s2, long bs2, const int* map, void* out, long bso, int Bt, int n, int C, hipStream_t st);
/* Two row sets through one map: oa[b][p] = sa[b][map[p]], ob[b][p] = sb[b][map[p]] (map NULL = identity; sb / ob NULL = one set).  The VidToMe chain moves a
 * token block and its cosine-normalised rows (the matching metric, merge.py:86-87) together: the survivors of the local merge go straight into their slot of the
 * global match's [src | dst] block and the new bank (patch.py:76-80) into the block of the chunk that will meet it, so no concatenation copy
 * (patch.py:62-70 `torch.cat`) and no second normalisation exist.  bs* / bo*: elements between batch entries. */
int tcl_gather_rows_pair_f16(const void* sa, long bsa, const void* sb, long bsb, const int* map, void* oa, long boa, void* ob, long bob, int Bt, int n, int C,
                             hipStream_t st);
/* h[b][i] += y[b][map[i]]  (unmerge + residual add, patch.py:178-179). */
int tcl_gather_add_rows_f16(void* h, long bsh, const void* y, long bsy, const int* map, int Bt, int n, int C, hipStream_t st);

/* ---- stage-2 input producer ---- */
/* get_soft_mask_bwds  utils/flow_utils.py:40-54: img [N,3,H,W], fwd/past flows [N,2,H,W] -> mask [N,1,H,W];
 * thr_abs = org_images.max() * diff_threshold. */
int tcl_soft_mask_bwds(const float* img, const float* fwd, const float* past, int N, int H, int W, float alpha, float beta, float thr_abs,
                       float* mask, hipStream_t st);
/* get_flowid  utils/flow_utils.py:56-93: ids int32 [N,H,W]; *last_id (device int) ends as the number of ids;
 * thr_abs = frames.max() * rgb_threshold.  Conflicting writes: the largest source index wins (the reference's CPU order). */
size_t tcl_flowid_workspace_bytes(int H, int W);
int tcl_flowid(const float* frames, const float* fwd_flows, const float* masks, int N, int H, int W, float thr_abs, int* ids, int* last_id,
               void* ws, hipStream_t st);

/* Measurement aid (bench.py roofline leg): bracket every flash-kernel launch (head_dim == dfilter, 0 = all) with HIP events
 * on its own stream; _end returns the summed kernel time, the algorithmic FLOPs 4*B*H*Tq*Tk*d and the launch count (a launch =
 * one attention call: the speculative kernel and the gated exact kernel behind it are timed together). */
int tcl_flash_profile_begin(int dfilter);
int tcl_flash_profile_end(double* total_ms, double* total_flops, long* launches);
int tcl_flash_profile_shape(int* shape4);        /* (B, H, Tq, Tk) of the largest launch profiled since _begin */
/* The same for the other two matrix-pipe consumers of path 1 (bench.py `roofline_gemm`, `roofline_match`): mask bit 0 = every
 * tcl_gemm_f16 / tcl_conv3x3_f16 / tcl_ln_gemm_f16 call (work = 2 M N K FLOP, the conv and Linear layers of SURVEY A9), bit 1 = every
 * tcl_tome_match*_f16 call (work = 2 n_src n_dst C Bt FLOP, the score GEMM of merge.py:84-108 / :389-421; the call's threshold and
 * map kernels are inside the bracket).  _end(cls = bit index) synchronises that class's events and switches it off. */
int tcl_prof_begin(int mask);
int tcl_prof_end(int cls, double* total_ms, double* total_work, long* launches);

/* Split-KV attention for ONE entry, head_dim 128 (round 6: the MemFlowNet memory read, memory_manager_skflow.py:44-69 -- one head, 14 400 queries at 1280x720
 * are 114 blocks for 256 CUs).  The keys are cut into nsplit chunks that run as batch entries of the flash kernel; each writes its normalised partial output and
 * log2 of its softmax denominator, a merge kernel joins them.  q [Tq, ldq], k [Tk, ldk], v [Tk, ldv] (head h at column h*d), o [Tq, ldo]; Tk % (64 nsplit) == 0. */
size_t tcl_attention_splitkv_workspace_bytes(int nsplit, int H, int Tq, int Tk, int d);
int tcl_attention_splitkv_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int H, int Tq, int Tk, int d,
                              float scale, int nsplit, void* ws, hipStream_t st);

/* ---- MemFlowNet correlation lookup (utils/evaluation/memflow/core/Networks/MemFlowNet/corr.py:74-120 CorrBlock,
 * computed on demand like the reference's unused alt_cuda_corr extension -- the all-pairs volume never exists).  f32, NHWC feature maps.
 * tcl_avgpool2_nhwc_f32: F.avg_pool2d(2, 2) (floor on odd sizes) -- builds the fmap2 pyramid level by level.
 * tcl_corr_lookup_f32: fmap1 [B,H,W,D]; fmap2_levels / level_h / level_w: HOST arrays of num_levels (<= 4) device pointers [B,h_l,w_l,D] and
 * their sizes; coords [B,2,H,W] (channel 0 = x, 1 = y, in fmap2 pixels of level 0); out = [B, L*(2r+1)^2, H, W] when out_nchw (the
 * reference's layout) else [B,H,W,L*(2r+1)^2]; channel = level*(2r+1)^2 + a*(2r+1) + b with x offset a-r and y offset b-r (the
 * reference's meshgrid(dy, dx) order), scaled by 1/sqrt(D).  D % 64 == 0, D <= 512, radius <= 5. */
int tcl_avgpool2_nhwc_f32(const float* x, float* y, int B, int H, int W, int D, hipStream_t st);
/* as tcl_corr_lookup_f32, written as f16 rows [B*H*W, ld] (channels [0, L*(2r+1)^2), ld >= that; padding channels are left untouched) */
int tcl_corr_lookup_rows_f16(const float* fmap1, const float* const* fmap2_levels, const int* level_h, const int* level_w, int num_levels,
                             const float* coords, void* out_rows, int ld, int B, int H, int W, int D, int radius, hipStream_t st);
int tcl_corr_lookup_f32(const float* fmap1, const float* const* fmap2_levels, const int* level_h, const int* level_w, int num_levels,
                        const float* coords, float* out, int B, int H, int W, int D, int radius, int out_nchw, hipStream_t st);
/* tcl_corr_lookup_rows_f16 for ONE entry and radius 4 with the neighbour rows shared by 8 x 8 pixel tiles (round 6; the windows of a tile's pixels overlap
 * almost completely under a smooth flow): fmap1_h / fmap2_levels_h = the same maps in f16 (level 0 is the encoder's f16 output itself, the pooled levels are
 * rounded once); a tile whose windows span more than 512 points is computed by the per-pixel f32 kernel from fmap1 / fmap2_levels instead (same call).
 * tile_flags: num_levels * ceil(H/8) * ceil(W/8) ints of scratch, rewritten by every call (1 = that tile took the per-pixel route). */
int tcl_corr_lookup_rows_tiled_f16(const void* fmap1_h, const void* const* fmap2_levels_h, const float* fmap1, const float* const* fmap2_levels,
                                   const int* level_h, const int* level_w, int num_levels, const float* coords, void* out_rows, int ld, int H, int W, int D,
                                   int radius, int* tile_flags, hipStream_t st);

/* ---- BriaRMBG-1.4 matting (briarmbg.py, generate.py:147-167).  f32 NCHW.
 * tcl_conv3x3_direct_f32: Conv2d(k=3, padding=dilation, dilation, stride 1|2) over x = cat([x1, x2], channel) (x2 may be NULL, C2 = 0)
 *   with w_t = weight.reshape(Cout, Cin*9).T contiguous ([Cin*9, Cout], tap index ky*3+kx), then y = conv*scale[oc] + shift[oc]
 *   (eval BatchNorm and the conv bias folded in), optional ReLU, then + resid (REBNCONV briarmbg.py:11-25; RSU tail `hx1d + hxin`).
 * tcl_maxpool2_ceil_f32: MaxPool2d(2, stride=2, ceil_mode=True) over BC planes.
 * tcl_resize_bilinear_f32: F.interpolate(size=(Ho,Wo), mode="bilinear") (align_corners=False), result * mul, optional sigmoid / clamp to [0,1]. */
int tcl_conv3x3_direct_f32(const float* x1, int C1, const float* x2, int C2, const float* w_t, const float* scale, const float* shift,
                           const float* resid, float* y, int B, int H, int W, int Cout, int dilation, int stride, int relu, hipStream_t st);
int tcl_maxpool2_ceil_f32(const float* x, float* y, int BC, int H, int W, hipStream_t st);
int tcl_resize_bilinear_f32(const float* x, float* y, int BC, int H, int W, int Ho, int Wo, float mul, int sigmoid, int clamp01, hipStream_t st);

/* ---- MemFlowNet encoders (core/Networks/MemFlowNet/cnn.py:124-216 BasicEncoder) -- the pieces beside tcl_gemm_f16 / tcl_conv3x3_f16.
 * tcl_conv7x7s2_c3_f16: Conv2d(3, 64, 7, stride 2, padding 3) on x [B,3,H,W] f32 NCHW -> y [B,Ho,Wo,64] f16 NHWC; w_t [147,64] f32 with
 *   row c*49 + ky*7 + kx (an eval BatchNorm may be folded into w_t / bias), optional ReLU.
 * tcl_instnorm_f16: InstanceNorm2d (affine=False, biased variance, eps) over [B,HW,C] f16 NHWC, optional ReLU; deterministic.
 * tcl_add_act_f16: y = act(a + b), act 0 none / 3 ReLU / 4 GELU(erf); n elements (n % 8 == 0).
 * tcl_subsample2_nhwc_f16: y[b][i][j] = x[b][2i][2j] (the pixel selection of a stride-2 1x1 convolution). */
int tcl_conv7x7s2_c3_f16(const float* x, const float* w_t, const float* bias, void* y, int B, int H, int W, int relu, hipStream_t st);
size_t tcl_instnorm_workspace_bytes(int B, int C);
int tcl_instnorm_f16(const void* x, void* y, int B, int HW, int C, float eps, int relu, void* ws, hipStream_t st);
int tcl_add_act_f16(const void* a, const void* b, void* y, long n, int act, hipStream_t st);
int tcl_subsample2_nhwc_f16(const void* x, void* y, int B, int H, int W, int C, hipStream_t st);

/* ---- MemFlowNet update block (core/Networks/MemFlowNet/sk2.py, MemFlow.py:172-183) -- glue beside the GEMMs, f16 NHWC rows unless noted.
 * tcl_dwconv_gelu_f16: y = gelu(x + depthwise_kxk(x) + bias) (PCBlock4_Deep_nopool_res, sk2.py:26-27); w [k*k, C] f16 (tap-major), k in {1, 7, 15} (the sizes sk2.py:201-202 uses).
 * tcl_nchw_f32_to_rows_f16 / tcl_rows_f16_to_nchw_f32: move Cs channels between an f32 NCHW tensor [B,Cs,P] and channels [c0, c0+Cs) of f16
 *   rows [B*P, ld] (zero_rest clears the other channels; the reverse computes out = alpha*out + beta*value, alpha 0 = overwrite).
 * tcl_axpy_f16: y = a + s*b.   tcl_upsample_flow_f32: MemFlowNet.upsample_flow -- softmax over the 9 taps of mask_scale*mask (rows
 *   [B*h*w, ldm], channel tap*64 + i*8 + j) applied to the 3x3 neighbourhood of 8*flow [B,2,h,w] -> [B,2,8h,8w]. */
/* tcl_context_split_f16: MemFlowNet.encode_context (MemFlow.py:112-115): c rows [P,256] -> net = tanh(c[:, :128]), inp = relu(c[:, 128:]). */
int tcl_context_split_f16(const void* c, void* net, void* inp, long P, hipStream_t st);
int tcl_dwconv_gelu_f16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int C, int k, hipStream_t st);
int tcl_nchw_f32_to_rows_f16(const float* x, void* y, int B, int Cs, int P, int ld, int c0, int zero_rest, hipStream_t st);
int tcl_rows_f16_to_nchw_f32(const void* x, float* y, int B, int Cs, int P, int ld, int c0, float alpha, float beta, hipStream_t st);
int tcl_axpy_f16(const void* a, const void* b, float s, void* y, long n, hipStream_t st);
int tcl_upsample_flow_f32(const float* flow, const void* mask, int ldm, float mask_scale, float* up, int B, int h, int w, hipStream_t st);

#ifdef __cplusplus
}
#endif
#endif
