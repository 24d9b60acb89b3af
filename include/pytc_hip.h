/* pytc_hip.h -- C ABI of the MI355X (gfx950) engine for the PyTorch Connectomics hot path.
 *
 * Every entry point takes raw device pointers + shapes + a hipStream_t (passed as void*),
 * launches asynchronously on that stream and returns an int status (0 = ok).  No torch
 * types, no ownership transfer: the caller allocates every buffer, including workspaces.
 * Re-entrant per stream; the only global state is a thread-local last-error string.
 *
 * The reference (PyTorch Connectomics, 100 % Python) has no FFI of its own for this path:
 * the boundary is the two Python plug-in interfaces of SURVEY.md section 8(b).  Each entry
 * below names the reference call site(s) (path:line under /root/reference/connectomics/)
 * whose arithmetic it replaces; INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Layouts: activations are NDHWC ("channels last 3d"), dtype PYTC_F32 or PYTC_BF16;
 *          parameters are fp32 in the packed layouts documented per function;
 *          sliding-window volumes / accumulators are fp32 [C][Z][Y][X] (== NCDHW, N = 1).
 */
#ifndef PYTC_HIP_H
#define PYTC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): pytc_mlp_args.per_sample (added in round 4 without a bump: ADVICE r04), pytc_dwconv3d_fwd with y = NULL, pytc_dwmix_*.
 * 4 (round 6): pytc_reduce_item.out_t (in the struct's former padding: a caller built against 3 may pass garbage there),
 * pytc_copy_zero_front, pytc_dwconv3d_bwd_data_add, pytc_pw_wgrad_groupnorm leaves the per-sample terms in the partials region at sps == 1.
 * Bumped whenever a struct layout or the meaning of an argument changes; _native.py refuses a library of another version. */
#define PYTC_ABI_VERSION 4

#define PYTC_OK 0
#define PYTC_ERR_INVALID 1     /* bad argument (shape, dtype, alignment) */
#define PYTC_ERR_HIP 2         /* a HIP runtime call / launch failed */
#define PYTC_ERR_UNSUPPORTED 3 /* valid request this build has no kernel for */

#define PYTC_F32 0
#define PYTC_BF16 1

/* window "view" bits (test-time-augmentation as index math, inference/tta.py:712-719,1026-1060) */
#define PYTC_VIEW_FLIP_Z 1
#define PYTC_VIEW_FLIP_Y 2
#define PYTC_VIEW_FLIP_X 4
#define PYTC_VIEW_SWAP_YX 8
/* round 6: quarter turns in the planes that contain z (inference/tta_combinations.py:90-119 accepts any plane): the window axes of the
 * plane are exchanged after the flips; at most ONE swap bit per view, and the two exchanged axes of the window must have equal length */
#define PYTC_VIEW_SWAP_ZY 16
#define PYTC_VIEW_SWAP_ZX 32

/* padding modes of inference/window.py:464-527 */
#define PYTC_PAD_CONSTANT 0
#define PYTC_PAD_REFLECT 1
#define PYTC_PAD_REPLICATE 2
#define PYTC_PAD_CIRCULAR 3

/* blending-map combine rules (inference/window.py:137-243) */
#define PYTC_BLEND_PRODUCT 0 /* w = max(max((wz*wy)*wx, FLT_MIN), floor)  (constant / bump) */
#define PYTC_BLEND_MIN 1     /* w = min(min(wz, wy), wx)                    (distance transform) */

/* activations */
#define PYTC_ACT_NONE 0
#define PYTC_ACT_SIGMOID 1
#define PYTC_ACT_TANH 2
#define PYTC_ACT_GELU 3

/* pointwise epilogue modes */
#define PYTC_RES_NONE 0
#define PYTC_RES_ADD 1        /* y = f(x) + R[row]                     (MedNeXt block / down block) */
#define PYTC_RES_UPSAMPLE 2   /* MedNeXt up block: front-pad + transposed 1x1 residual + skip     */
#define PYTC_RES_GELU_BWD 3   /* y = f(x) * gelu'(R[row]): GELU backward fused into the data-gradient GEMM */
#define PYTC_RES_NORM_BWD 4   /* y = A[n][o]*f(x) + B[n][o]*R[row][o] + C[n][o], (A, B, C) = res_bias [N][3][C_out]: the GroupNorm backward
                               * apply pass fused into the data-gradient GEMM that produces its operand (w_paired kernel only).
                               * With (Di, Hi, Wi) != 0 (round 6, up blocks): the rows are the padded grid of a transposed conv's output,
                               * y is the compact (Di - 1, Hi - 1, Wi - 1) grid -- rows on a front face are not written */

int pytc_abi_version(void);
const char* pytc_last_error(void);
/* fills cu_count / lds_bytes_per_cu / gcn arch name of `device`; returns status */
int pytc_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len);
/* integer tuning knobs (kernel-variant selection for A/B measurements; defaults are the tuned ones) */
int pytc_set_tuning(const char* key, int value);

/* ---------------------------------------------------------------- sliding window --------- */

/* Patch gather + boundary pad.  Replaces _extract_padded_patch_batch
 * (inference/window.py:464-527) and the view flips of inference/tta.py:1021-1030.
 *   vol    fp32 [C][Z][Y][X] (device)
 *   starts host int32 [B][3] window origins (may be negative / overhang: padded by pad_mode)
 *   out    [B][rz][ry][rx][C] in out_dtype
 * B <= 64. */
int pytc_gather_windows(const float* vol, int C, int Z, int Y, int X, const int32_t* starts, int B,
                        int rz, int ry, int rx, int view, int pad_mode, float cval, void* out,
                        int out_dtype, void* stream);

/* Overlap-add of B window predictions, one launch per window IN ORDER (so the fp32 sum order
 * is the reference's).  Replaces EagerSlidingWindowEngine._accumulate
 * (inference/window.py:648-655), lazy.py:1216-1227, tta.py:1161-1186.
 *   pred   [B][rz][ry][rx][C] in pred_dtype (network output, NDHWC)
 *   wz/wy/wx fp32 per-axis blending factors (device), combined by `combine`, floored by `floor_w`
 *   value  fp32 [C][Z][Y][X]  +=  pred * w      weight fp32 [Z][Y][X] += w  (if weight != NULL)
 * Voxels falling outside [0,Z)x[0,Y)x[0,X) are skipped (this is how region / chunk accumulators clip the
 * global window grid, inference/lazy.py:1069-1099).  `border` zeroes the outer voxels of the window map
 * (host int32[3]; apply_border_mask, inference/window.py:297-319). */
int pytc_blend_accumulate(const void* pred, int pred_dtype, int B, const int32_t* starts, int rz,
                          int ry, int rx, int C, int view, const float* wz, const float* wy,
                          const float* wx, int combine, float floor_w, const int32_t* border /*[3] or NULL*/,
                          float* value, float* weight, int Z, int Y, int X, void* stream);

/* Affinity-aware overlap-add (inference/tta_affinity.py:350-393 `invert_view` + tta.py:1108-1186): like
 * pytc_blend_accumulate, but output channel d takes prediction channel chan_src[d] displaced by chan_shift[d] (host
 * int32 [C] / [C][3], window-local canonical coordinates): the value predicted at q lands at p = q + shift and is
 * weighted by the blending map at p; p outside the window is dropped (the wrapped face).  C <= 32.  `weight` (may be
 * NULL) receives the un-shifted map as in pytc_blend_accumulate. */
int pytc_blend_accumulate_mapped(const void* pred, int pred_dtype, int B, const int32_t* starts, int rz, int ry,
                                 int rx, int C, int view, const float* wz, const float* wy, const float* wx,
                                 int combine, float floor_w, const int32_t* border, const int32_t* chan_src,
                                 const int32_t* chan_shift, float* value, float* weight, int Z, int Y, int X,
                                 void* stream);
/* weight += blending map restricted to the positions that a window displaced by `shift` (host int32[3]) covers
 * (valid_slices_for_shift, tta_affinity.py:100-119; the per-shift weight accumulators of tta.py:1121-1140). */
int pytc_blend_weight_shifted(int B, const int32_t* starts, int rz, int ry, int rx, const float* wz, const float* wy,
                              const float* wx, int combine, float floor_w, const int32_t* border,
                              const int32_t* shift, float* weight, int Z, int Y, int X, void* stream);
/* value[i] = weight[i] > 0 ? value[i] / weight[i] : 0, in place (tta.py:1238-1244: partial channels are normalised by
 * their own coverage, without the 1e-4 clamp; coverage = weight > 0 is their validity). */
int pytc_normalize_covered(float* value, const float* weight, int64_t n, void* stream);
/* Validity-aware running statistics over TTA views (inference/tta_ensemble.py:122-165): where cover[i] > 0 (cover NULL
 * = everywhere): mode 0 stat += x, mode 1 stat = min(stat, x), mode 2 max; count += 1.  Finalize (:205-210): mode 0
 * out = stat / count, else out = stat; the caller rejects count == 0. */
int pytc_ensemble_update_masked(float* stat, float* count, const float* x, const float* cover, int64_t n, int mode,
                                void* stream);
int pytc_ensemble_finalize_masked(const float* stat, const float* count, float* out, int64_t n, int mode, void* stream);

/* value[c][i] = act(value[c][i] / max(weight[i], clamp)), in place.  Replaces
 * normalize_weighted_accumulator (inference/window.py:275-294) + the sigmoid/tanh of
 * apply_preprocessing (inference/tta.py:312-402). */
int pytc_blend_finalize(float* value, const float* weight, int C, int64_t nvox, float clamp,
                        int act, void* stream);

/* In-place activation of channels [c0, c1) of a fp32 volume laid out [C][nvox] (channels_last = 0) or
 * [nvox][C] (channels_last = 1, window predictions): v = act(scale * v) with act in
 * {NONE, SIGMOID, TANH} or PYTC_ACT_SOFTMAX (across the channel group, scale ignored).  Replaces
 * TTAPredictor.apply_preprocessing (inference/tta.py:312-402: sigmoid / scale_sigmoid:<t> / tanh / softmax). */
#define PYTC_ACT_SOFTMAX 4
int pytc_channel_activation(float* value, int C, int64_t nvox, int channels_last, int c0, int c1, int act,
                            float scale, void* stream);

/* out = running ensemble update over TTA views (inference/tta_ensemble.py:85-101):
 * mode 0 mean: acc += (x - acc) / count ; mode 1 min ; mode 2 max.  n elements fp32. */
int pytc_ensemble_update(float* acc, const float* x, int64_t n, int mode, int count, void* stream);

/* Prediction / storage dtype transform on device (inference/output.py:150-243: _apply_intensity_transform,
 * _convert_intensity_dtype): y = cast(clip(x * scale)) with numpy semantics -- integer targets clip to the type's range
 * first and truncate toward zero in the cast; float targets only round.  scale <= 0 or == 1 leaves values unscaled
 * (the reference treats a negative scale as "disabled").  x: fp32 [n]; y: n elements of the target type. */
#define PYTC_ST_U8 0
#define PYTC_ST_I8 1
#define PYTC_ST_U16 2
#define PYTC_ST_I16 3
#define PYTC_ST_I32 4
#define PYTC_ST_F16 5
#define PYTC_ST_F32 6
int pytc_scale_cast(const float* x, void* y, int64_t n, float scale, int target, void* stream);

/* ---------------------------------------------------------------- disk-backed test volumes -- */

/* The transform half of the reference's LazyVolumeAccessor (inference/lazy.py:456-917: `_read_raw_crop` :691-717,
 * `_read_transformed_bbox` :719-780, `_read_padded_inner_region` :782-850) as ONE gather over the raw storage box:
 *   out[c][z][y][x] = blend over {i0, i1} per axis of raw[c*s_c + iz*s_z + iy*s_y + ix*s_x]
 * raw: the storage bytes of the box (device), element type `raw_dtype`; strides_czyx (host, elements): where the CHANNEL and
 * the three LOGICAL axes (after `val_transpose`) live in the box, so channel layouts and the transpose cost nothing;
 * tab_i0 / tab_i1 / tab_f (device, nz + ny + nx entries each, z | y | x back to back): per output index the two box-local raw
 * indices it reads and the weight of the second (0: nearest / no resize; i0 < 0: outside a constant context pad -> 0) --
 * resize (nearest for labels and masks, trilinear align_corners=True for images) and context padding (constant / reflect /
 * edge) are folded into the tables by the host (inference/lazy_accessor.py).  out: fp32 (C, nz, ny, nx). */
#define PYTC_RAW_U8 0
#define PYTC_RAW_I8 1
#define PYTC_RAW_U16 2
#define PYTC_RAW_I16 3
#define PYTC_RAW_U32 4
#define PYTC_RAW_I32 5
#define PYTC_RAW_F32 6
#define PYTC_RAW_F64 7
int pytc_resample_region(const void* raw, int raw_dtype, const int64_t* strides_czyx, int C, const int32_t* tab_i0,
                         const int32_t* tab_i1, const float* tab_f, const int32_t* dims_zyx, float* out, void* stream);

/* The per-window finishing of `read_patch` (lazy.py:896-904) + `smart_normalize` (data/augmentation/augment_ops.py:552-611) on a
 * batch of gathered windows x fp32 (B, n), in place: optional binarisation (v > threshold), optional clipping to per-window bounds
 * clip (B, 2) (the percentile bounds), then mode: NONE | ZSCORE ((v - mean) / std when std > 1e-8) | MINMAX ((v - min) / (max - min)
 * when max > min) | DIVIDE (v / divide).  Statistics are those of THAT window after binarise / clip, two-stage fixed-order reduction
 * in fp64.  workspace: pytc_window_normalize_ws_elems(B, n) doubles (statistics modes only). */
#define PYTC_NORM_NONE 0
#define PYTC_NORM_ZSCORE 1
#define PYTC_NORM_MINMAX 2
#define PYTC_NORM_DIVIDE 3
int64_t pytc_window_normalize_ws_elems(int B, int64_t n);
int pytc_window_normalize(float* x, int B, int64_t n, int mode, int binarize, float threshold, float divide, const float* clip,
                          double* workspace, void* stream);

/* ---------------------------------------------------------------- depthwise conv ---------- */

/* number of per-sample partial-statistics slots pytc_dwconv3d_fwd / pytc_dwconvT3d_fwd write for
 * this problem (batch N, INPUT dims); -1 when the channel count is unsupported */
int pytc_dwconv3d_stat_slots(int N, int D, int H, int W, int C, int K, int stride, int dtype,
                             int transposed);

/* which kernel family pytc_dwconv3d_fwd / pytc_dwconvT3d_fwd runs for this problem (measurement bookkeeping only:
 * bench.py groups its per-launch timings by device kernel): 0 direct, 1 K=3/5/7 gather, 2 x-block, 3 z-march,
 * 4 transposed 2x2x2-cell, 5 transposed direct, 6 z-march on the matrix cores (bf16 forward),
 * 7 transposed tile form (bf16, C = 64 / 128), 8 stride-2 z-march (bf16, C = 32 / 64); -1 unsupported channel count */
int pytc_dwconv3d_kernel_variant(int N, int D, int H, int W, int C, int K, int stride, int dtype, int transposed);

/* Depthwise Conv3d (groups == C), kernel K^3 (3/5/7), padding K/2, stride 1 or 2, fused with the
 * per-(n,c) sum / sum-of-squares of the OUTPUT that the following GroupNorm(C,C) needs.
 * Replaces MedNeXtBlock.conv1 (+ first half of .norm) -- external nnunet_mednext, attribute
 * contract at models/architectures/mednext_models.py:104-117.
 *   x [N][D][H][W][C]   y [N][Do][Ho][Wo][C]   (dtype)      Do = (D + 2*(K/2) - K)/stride + 1
 *   w fp32 [K*K*K][C] (tap-major: w[(kz*K+ky)*K+kx][c] = torch_weight[c][0][kz][ky][kx])
 *   bias fp32 [C] or NULL
 *   stats fp32 [N][slots][2][C] (sum, sumsq per slot) or NULL
 * Arithmetic: fp32 accumulation of fp32 taps, except the bf16 / K = 3 / stride 1 / C % 32 == 0 launches on planes of >= 8 x 8 voxels, which
 * run on the matrix cores (csrc/dwconv_mfma_kernels.hip): the taps enter as bf16 (round to nearest even -- what torch.autocast hands the
 * reference's Conv3d), products are exact, accumulation is fp32; knob dwconv_mfma_variant bit 0 adds the low halves of the taps (16-bit
 * mantissa; always on for planes below 16 voxels), knob dwconv_mfma = 0 restores the VALU kernels (fp32 taps, nine-tap f16 partial sums).
 * The statistics are those of the STORED (rounded) tensor and their slot count does not depend on N.
 * y = NULL (statistics must be given): the statistics-only pass of the matrix-core launches -- the same products, roundings and partial
 * sums, bit-identical statistics, nothing stored (first pass of pytc_dwmix_fwd); other kernel families refuse a NULL output. */
int pytc_dwconv3d_fwd(const void* x, void* y, const float* w, const float* bias, float* stats,
                      int N, int D, int H, int W, int C, int K, int stride, int dtype,
                      void* stream);

/* pytc_dwconv3d_fwd for operands of arbitrary magnitude -- the training backward convolves GRADIENT tensors (|v| ~ 1e-7 under a
 * mean-reduced loss), which the packed-f16 in-plane sums of the bf16 z-march kernel cannot represent: this entry keeps fp32 taps
 * and fp32 partial sums on every path.  Same arguments and semantics otherwise. */
int pytc_dwconv3d_fwd_wide(const void* x, void* y, const float* w, const float* bias, float* stats, int N, int D, int H,
                           int W, int C, int K, int stride, int dtype, void* stream);

/* y = pytc_dwconv3d_fwd(x) + res (res laid out like y, no statistics): the data gradient of a residual MedNeXt block,
 * dx = conv_reversed_taps(dt) + dy, without the separate add pass.  bf16, z-march shapes only (K = 3, stride 1, C % 32 == 0,
 * D >= 8, H, W >= 16): pytc_dwconv3d_res_supported; other shapes: convolve, then pytc_add_inplace.  The sum is formed on the
 * fp32 accumulator and rounded once (the two-step form rounds the convolution first); fp32 taps / partial sums always
 * (gradient operands, see pytc_dwconv3d_fwd_wide).  Replaces the autograd accumulation of the residual
 * branch (MedNeXtBlock.forward: x + conv path; mednext attribute contract at mednext_models.py:104-117). */
int pytc_dwconv3d_res_supported(int D, int H, int W, int C, int K, int stride, int dtype);

/* Fused MedNeXt residual block (round 5; kind "block": conv1 -> norm -> conv2 -> act -> conv3 -> + x of the external nnunet_mednext
 * MedNeXtBlock.forward, contract at mednext_models.py:99-126) whose depthwise output never reaches HBM -- SURVEY.md 8(d)'s byte floor
 * "read x twice, write y once":
 *   1. pytc_dwconv3d_fwd(x, y = NULL, taps, dw_bias, stats, ...)      statistics of the depthwise output (nothing stored)
 *   2. pytc_groupnorm_fold_mlp(stats, ...) -> w2n, b2n                GroupNorm folded into per-sample expand operands
 *   3. pytc_dwmix_fwd(...)                                            the depthwise conv re-formed plane by plane in LDS by the same
 *      matrix-core kernel, each plane's 64 positions x 32 channels handed through LDS to the channel mixer (expand -> packed-fp16 GELU
 *      -> fp16 projection -> + x when residual = 1), y [N][D][H][W][32] bf16.
 * x [N][D][H][W][32] bf16; taps [27][32], dw_bias [32] fp32; w2n = N paired bf16 images of [C_hid][32], b2n [N][C_hid]; w3_f16 =
 * pytc_pw_pack_weight_paired_f16 image of [32][C_hid], b3 [32].  head_w != NULL: pytc_pw_mlp_head_fwd's epilogue (head_y
 * [N][D*H*W][n_head] fp32; y may then be NULL).  Results are BIT-IDENTICAL to pytc_dwconv3d_fwd + pytc_pw_mlp_fwd (per_sample = 1,
 * w3_format = PYTC_W3_F16) / pytc_pw_mlp_head_fwd on the same operands.  C = C_out = 32, C_hid in {64, 96, 128}. */
int pytc_dwmix_supported(int D, int H, int W, int C, int C_hid, int C_out, int dtype);
int pytc_dwmix_fwd(const void* x, const float* taps, const float* dw_bias, const void* w2n, const float* b2n, const void* w3_f16,
                   const float* b3, int residual, void* y, const void* head_w, const float* head_b, float* head_y, int n_head, int N,
                   int D, int H, int W, int C, int C_hid, int C_out, int dtype, void* stream);
int pytc_dwconv3d_fwd_res(const void* x, const void* res, void* y, const float* w, const float* bias, int N, int D, int H,
                          int W, int C, int K, int stride, int dtype, void* stream);

/* Depthwise ConvTranspose3d (groups == C), kernel K^3, stride 2, padding K/2 -> Do = 2D-1.
 * Replaces MedNeXtUpBlock.conv1.  Output is written into a buffer of spatial size
 * (2D)x(2H)x(2W) at offset +1 on every axis (the block's later F.pad((1,0,1,0,1,0)) is thereby
 * free); the front faces of y are NOT touched.  w fp32 [K^3][C] with
 * w[(kz*K+ky)*K+kx][c] = torch_weight[c][0][kz][ky][kx].  stats as above, over (2D-1)^3. */
int pytc_dwconvT3d_fwd(const void* x, void* y, const float* w, const float* bias, float* stats,
                       int N, int D, int H, int W, int C, int K, int dtype, void* stream);

/* Reduce the partial statistics and emit the per-(n,c) affine of GroupNorm(num_groups=C):
 *   mean = S1/count, var = S2/count - mean^2 (biased), a = gamma * rsqrt(var+eps), b = beta - mean*a
 *   ab fp32 [N][2][C].  Replaces the statistics half of nn.GroupNorm in MedNeXtBlock.norm. */
int pytc_groupnorm_finalize(const float* stats, int slots, float count, const float* gamma,
                            const float* beta, float eps, float* ab, int N, int C, void* stream);

/* ---------------------------------------------------------------- pointwise (1x1x1) ------- */

/* Pack a PyTorch 1x1x1 conv weight [C_out][C_in] (fp32, row major) into the MFMA operand image
 * the GEMM kernels read.  `packed` must hold pytc_pw_packed_elems(C_out, C_in) elements of
 * `dtype`.  transposed != 0 means the source is [C_in][C_out] (ConvTranspose3d layout). */
int64_t pytc_pw_packed_elems(int C_out, int C_in, int dtype);
int pytc_pw_pack_weight(const float* w, int C_out, int C_in, int transposed, void* packed,
                        int dtype, void* stream);

/* Generic pointwise convolution  y[r][o] = act( sum_k f(x[src(r)][k]) * W[o][k] + bias[o] ) (+ res)
 *   f(x) = a[n][k]*x + b[n][k] when ab != NULL (GroupNorm apply), identity otherwise
 *   rows = N * rows_per_sample; x is [N*rows_per_sample_in][C_in] (in_dtype)
 *   gather: 0 dense (src(r) = r); 2 = stride-2 spatial subsample of an input grid
 *           (Di,Hi,Wi) -> (Do,Ho,Wo) = ceil(./2)   (MedNeXtDownBlock.res_conv, k=1 stride 2)
 *   res_mode PYTC_RES_ADD adds res[r][o] (out dtype layout);
 *   Replaces stem / conv2 / conv3 / res_conv / out_0 / task-head 1x1 convs
 *   (mednext_models.py:120,169-173,188; rsunet.py:249,388). */
typedef struct {
  const void* x;        /* input activations */
  const void* w_packed; /* from pytc_pw_pack_weight */
  const float* bias;    /* [C_out] or NULL */
  const float* ab;      /* [N][2][C_in] or NULL */
  const void* res;      /* residual or NULL */
  void* y;              /* output */
  int N;                /* samples */
  int64_t rows_per_sample; /* OUTPUT rows per sample */
  int C_in, C_out;
  int in_dtype, out_dtype, w_dtype;
  int act;              /* PYTC_ACT_* applied before the residual */
  int res_mode;         /* PYTC_RES_* */
  int gather;           /* 0 or 2 */
  int Di, Hi, Wi;       /* input grid (gather == 2) or output grid (RES_UPSAMPLE) */
  const void* res_low;  /* RES_UPSAMPLE: low-res residual [N][Di/2..][C_out] */
  const float* res_bias;/* RES_UPSAMPLE: bias of the transposed 1x1 residual conv */
  int pre_act;          /* PYTC_ACT_NONE or PYTC_ACT_GELU applied to f(x) BEFORE the GEMM (the stored tensor is the
                         * pre-activation; training keeps only that one copy of the expanded tensor) */
  int w_paired;         /* 0: w_packed from pytc_pw_pack_weight; 1: from pytc_pw_pack_weight_paired (bf16) -- selects
                         * the 16-byte-store kernel; allowed only when pytc_pw_conv_paired_supported(a) != 0;
                         * 2 (round 6): w_packed is the plain row-major bf16 matrix [C_out][C_in] (the conv weight cast, or its transpose
                         * for a data gradient) and the launch is the LDS-tiled GEMM of pytc_pw_gemm_fwd with every prologue (ab, pre_act)
                         * and epilogue (res_mode incl. PYTC_RES_GELU_BWD / PYTC_RES_NORM_BWD) of the paired-row kernel: the deep levels
                         * of the training step; allowed only when pytc_pw_conv_rowmajor_supported(a) != 0 */
} pytc_pw_args;

int pytc_pw_conv_fwd(const pytc_pw_args* a, void* stream);
/* 1 when the shape / dtypes / modes in `a` are covered by the paired-row kernel (pointers are not inspected) */
int pytc_pw_conv_paired_supported(const pytc_pw_args* a);
/* 1 when they are covered by the row-major LDS-tiled GEMM (w_paired = 2): bf16, C_in % 64 == 0, C_out % 128 == 0, no gather, no activation */
int pytc_pw_conv_rowmajor_supported(const pytc_pw_args* a);

/* Fused MedNeXt channel mixer (bf16 activations):
 *     y = W3 * gelu( W2 * (a*t + b) + b2 ) + b3   (+ residual per res_mode, as in pytc_pw_conv_fwd)
 * One launch for norm-apply, conv2, GELU, conv3 and the residual add of a MedNeXt block / down block /
 * up block; the expanded tensor stays in registers.  Replaces MedNeXtBlock.{norm (apply), conv2, act,
 * conv3} + the block residual (external nnunet_mednext; contract at mednext_models.py:104-126).
 * Weights must be packed with pytc_pw_pack_weight_paired (bf16).  Supported channel triples are
 * reported by pytc_pw_mlp_supported (multiples of 32 that occur in MedNeXt with 32 base channels);
 * other shapes go through two pytc_pw_conv_fwd calls. */
typedef struct {
  const void* t;          /* [N][rows][C_in] bf16: depthwise-conv output */
  const float* ab;        /* [N][2][C_in] GroupNorm affine from pytc_groupnorm_finalize */
  const void* w2_packed;  /* expand  C_in  -> C_hid */
  const float* b2;        /* [C_hid] */
  const void* w3_packed;  /* project C_hid -> C_out */
  const float* b3;        /* [C_out] */
  const void* res;        /* see PYTC_RES_* */
  const void* res_low;
  const float* res_bias;
  void* y;                /* [N][rows][C_out] bf16 */
  int N;
  int64_t rows_per_sample;
  int C_in, C_hid, C_out;
  int res_mode;
  int Di, Hi, Wi;         /* RES_UPSAMPLE: output grid */
  int w3_format;          /* PYTC_W3_BF16 (0): w3_packed from pytc_pw_pack_weight_paired; PYTC_W3_F16: from
                           * pytc_pw_pack_weight_paired_f16 -- the hidden activation then runs the packed-fp16 GELU and the
                           * projection the f16 MFMA (pytc_pw_mlp_fwd / pytc_pw_mlp_train_fwd only) */
  int per_sample;         /* 1: ab = NULL and the GroupNorm affine is already inside the expanding conv (pytc_groupnorm_fold_mlp):
                           * w2_packed = N images back to back, b2 = [N][C_hid]; the mixer reads t raw.  Inference entries only
                           * (pytc_pw_mlp_fwd, _head_fwd, _stemres_fwd) */
} pytc_mlp_args;
#define PYTC_W3_BF16 0
#define PYTC_W3_F16 1

/* Deep levels (round 4): the two 1x1x1 convs of a MedNeXt block as two LDS-tiled MFMA GEMM launches instead of the fused mixer, for
 * batches of fewer than ~30 k voxel rows (levels 3-4 of an 8-window batch), where the fused kernel's per-wave hidden loop leaves most
 * SIMDs idle.  y[r][o] = epi(sum_k f(x[r][k]) * w[o][k] + bias[o]): x [N * rows][C_in] bf16 (in_f16 = 0, optional GroupNorm affine
 * ab [N][2][C_in]) or fp16 (in_f16 = 1), w [C_out][C_in] ROW-MAJOR in the same 16-bit type (the conv weight as PyTorch stores it,
 * cast; no packed image), gelu_out = 1: y fp16 = packed-fp16 GELU of the accumulator (the hidden tensor), gelu_out = 0: y bf16 with
 * res / res_low / res_bias / res_mode / (Di, Hi, Wi) as pytc_pw_mlp_fwd.  A block computed as gemm(gelu_out = 1) -> gemm(in_f16 = 1)
 * is bit-identical to pytc_pw_mlp_fwd with w3_format = PYTC_W3_F16.  C_in % 64 == 0, C_out % 128 == 0.  Replaces conv2 / act / conv3 of
 * the external nnunet_mednext block (contract at mednext_models.py:99-126). */
int pytc_pw_gemm_supported(int C_in, int C_out);
int pytc_pw_gemm_fwd(const void* x, const void* w, const float* bias, const float* ab, void* y, int N, int64_t rows_per_sample,
                     int C_in, int C_out, int in_f16, int gelu_out, const void* res, const void* res_low, const float* res_bias,
                     int res_mode, int Di, int Hi, int Wi, void* stream);
int pytc_pw_mlp_supported(int C_in, int C_hid, int C_out);
int pytc_pw_pack_weight_paired(const float* w, int C_out, int C_in, int transposed, void* packed_bf16,
                               void* stream);
/* All per-step weight re-layouts of a model in one launch (training).  table_dev: device array [n_items][8] int64
 *   {src (fp32), dst, kind, C_out, C_in, aux, first_element, element_count}, rows sorted by first_element, contiguous:
 *   kind 0 / 1: pytc_pw_pack_weight_paired of a [C_out][C_in] / transposed source (aux = ceil(C_in/32));
 *   kind 2 / 3: the same as pytc_pw_pack_weight_paired_f16;
 *   kind 4 / 5: depthwise stencil [C][K^3] -> tap-major [K^3][C] fp32, as stored / reversed (aux = K^3, C_out = C).
 * Results are bit-identical to the individual pack calls. */
int pytc_pack_multi(const int64_t* table_dev, int n_items, int64_t total_elems, void* stream);
/* the same image in IEEE fp16 (round to nearest even) for pytc_mlp_args.w3_format = PYTC_W3_F16 */
int pytc_pw_pack_weight_paired_f16(const float* w, int C_out, int C_in, int transposed, void* packed_f16,
                                   void* stream);
int pytc_pw_mlp_fwd(const pytc_mlp_args* a, void* stream);

/* pytc_pw_mlp_fwd (w3_format = PYTC_W3_F16, forward only) for the mid-level shapes 64->128->64, 128->256->64 and 128->256->128, as a
 * PERSISTENT kernel with both weight images resident in LDS (csrc/pw_mlp_lds_kernels.hip): a workgroup stages the images once and
 * walks a contiguous share of the (sample, row tile) sequence; results are bit-identical to pytc_pw_mlp_fwd.  Same reference
 * boundary as pytc_pw_mlp_fwd (MedNeXtBlock.forward: norm -> conv2 -> act -> conv3 -> + x). */
int pytc_pw_mlp_lds_supported(int C_in, int C_hid, int C_out);
int pytc_pw_mlp_lds_fwd(const pytc_mlp_args* a, void* stream);
/* pytc_pw_mlp_fwd (w3_format = PYTC_W3_F16, forward only) for wide hidden layers whose weight images exceed LDS (C_in / C_out of
 * 64->128, 128->64, 128->128, 256->128; any C_hid that is a multiple of 32 up to 8192: MedNeXt-L's 128->1024->128, 256->2048->128,
 * 128->512->64, 64->512->128) -- csrc/pw_mlp_chunk_kernels.hip: the waves of a workgroup share each 32-wide hidden chunk's weight
 * fragments, streamed L2 -> LDS by DMA two chunks ahead; results are bit-identical to pytc_pw_mlp_fwd.  Same reference boundary. */
/* The mixer of a 32-channel residual block with a 32 -> 32 1x1x1 conv of its bf16-rounded output in the epilogue, z = bf16(W y + b) -- the input
 * projection of task heads behind the trunk's last block (reference: MedNeXtTaskHead.input_projection after dec_block_0, mednext_models.py:99-126;
 * here the block-diagonal merged projection).  proj_w: paired bf16 image of W (pytc_pw_pack_weight_paired, 32 x 32); y is written only when
 * store_y.  Per-sample (norm-folded) expand operands, fp16 projection image, C_hid in {64, 96, 128}. */
/* 1 when pytc_pw_mlp_fwd / pytc_pw_mlp_head_fwd / pytc_pw_mlp_stemres_fwd (ignore_res_mode = 1) will run these arguments on the DMA-prefetching
 * level-0 kernel (pw_mlp_dma_kernel), 0 for the one-tile-per-wave kernel: same results either way; for tools that name launches by device symbol. */
int pytc_pw_mlp_dma_applies(const pytc_mlp_args* a, int ignore_res_mode);
int pytc_pw_mlp_proj_supported(int C_in, int C_hid, int C_out, int C_proj);
int pytc_pw_mlp_proj_fwd(const pytc_mlp_args* a, const void* proj_w, const float* proj_b, void* z, int store_y, void* stream);
int pytc_pw_mlp_chunk_supported(int C_in, int C_hid, int C_out);
int pytc_pw_mlp_chunk_fwd(const pytc_mlp_args* a, void* stream);
/* GroupNorm finalize + fold into the mixer's expanding conv, one launch (replaces pytc_groupnorm_finalize in front of an
 * inference mixer: MedNeXtBlock.norm followed by conv2, external nnunet_mednext block; contract at mednext_models.py:99-126):
 *   a_n = gamma * rstd_n, b_n = beta - mean_n * a_n   from stats [N][slots][2][C] (fixed summation order per sample),
 *   w2n[n] = paired bf16 image of W2 * diag(a_n)   (pytc_pw_pack_weight_paired layout, N images back to back),
 *   b2n[n][o] = b2[o] + sum_k W2[o][k] * b_n[k]     (fp32),
 * so that W2n * t + b2n == W2 * (a*t + b) + b2 and the mixer's operand is the raw depthwise output.  w2 fp32 [C_hid][C]
 * row-major (the conv weight as PyTorch stores it), C in {32, 64, 128}, C_hid % 32 == 0, C_hid <= 512; ab_out [N][2][C] or NULL. */
int pytc_groupnorm_fold_mlp(const float* stats, int slots, float count, const float* gamma, const float* beta, float eps,
                            const float* w2, const float* b2, void* w2n, float* b2n, float* ab_out, int N, int C, int C_hid,
                            void* stream);
/* The same mixer with the network's 1x1x1 output projection (mednext OutBlock.conv_out, a transposed 1x1x1 conv on the
 * full-resolution features) in its epilogue: logits[o] = head_b[o] + sum_c head[o][c] * bf16(y[c]), o < n_head <= 16,
 * head_y [N][rows][n_head] fp32; head_w = the bf16 MFMA A-fragment image of the head [64 lanes][8]: lane (r, kb) holds
 * head[o = r][c = kb*8 .. kb*8+7] (zero rows for r >= n_head).  C_out must be 32 (pytc_pw_mlp_head_supported), residual
 * NONE or ADD.  store_y = 0 skips the 64 B / voxel block output (a->y may then be NULL). */
/* Training forward of the mixer: pytc_pw_mlp_fwd that also stores the hidden pre-activation W2*(a*t+b)+b2 as bf16
 * [N][rows][C_hid] (the GELU input the backward kernels differentiate); GELU is evaluated at that stored value. */
int pytc_pw_mlp_train_fwd(const pytc_mlp_args* a, void* hidden_pre, void* stream);
/* The training forward without the store (round 6): the hidden pre-activation is rounded to bf16 before the activation exactly as
 * pytc_pw_mlp_train_fwd rounds it -- y carries the same bits -- but is not written; the block's backward rebuilds it from the
 * depthwise output (pytc_mixer_bwd_rc).  128 of the 320 bytes per voxel the full-resolution 32 -> 64 -> 32 mixer moved. */
int pytc_pw_mlp_train_fwd_nostore(const pytc_mlp_args* a, void* stream);
/* Training backward of the mixer's data path in one launch: dX = W2^T ((W3^T dY) * GELU'(hidden_pre)); a->t = dY,
 * a->w2_packed = paired image of W3^T (C_out_fwd -> C_hid), a->w3_packed = paired image of W2^T (C_hid -> C_in_fwd),
 * a->ab = identity affine [N][2][C], a->b2 / a->b3 = zero vectors, a->res_mode = NONE, a->y = dX; d_hidden
 * [N][rows][C_hid] bf16 receives the intermediate (the operand of the expanding conv's weight gradient). */
int pytc_pw_mlp_bwd(const pytc_mlp_args* a, const void* hidden_pre, void* d_hidden, void* stream);
/* Fused MedNeXt UP block: the depthwise transposed 3x3x3 conv (stride 2, padding 1, written at the +1 offset of
 * MedNeXtUpBlock's F.pad((1,0,1,0,1,0))) is computed in the mixer's prologue from the LOW-resolution block input, so the
 * 2C-channel high-resolution tensor t never touches HBM: a->t = x_low [N][Di][Hi][Wi][C_in] bf16, a->Di/Hi/Wi = the
 * low-res grid, a->res = encoder skip [N][2Di][2Hi][2Wi][C_out], a->res_low / a->res_bias / a->y as for
 * PYTC_RES_UPSAMPLE, a->ab = GroupNorm affine of t (statistics: pytc_dwconvT3d_fwd with y = NULL, the statistics-only
 * mode of the same cell kernel); taps [27][C_in] fp32 tap-major; dw_bias [C_in] and a->res_bias [C_out] must be given
 * (zero vectors when the conv has no bias / the block has no residual conv).  Results are bit-identical to
 * pytc_dwconvT3d_fwd + pytc_pw_mlp_fwd(PYTC_RES_UPSAMPLE).  Replaces MedNeXtUpBlock.forward (external nnunet_mednext;
 * contract at mednext_models.py:104-126). */
int pytc_pw_mlp_up_supported(int C_in, int C_hid, int C_out);
int pytc_pw_mlp_up_fwd(const pytc_mlp_args* a, const float* taps, const float* dw_bias, void* stream);
int pytc_pw_mlp_head_supported(int C_in, int C_hid, int C_out);
/* The first block of the network (stem fused away, see pytc_stem_dwconv3d_fwd): the mixer's residual is the stem output
 * recomputed from the 1-channel input, res[c] = bf16(stem_w[c] * stem_x[voxel] + stem_b[c]); C_in = C_out = 32. */
int pytc_pw_mlp_stemres_fwd(const pytc_mlp_args* a, const float* stem_x, const float* stem_w, const float* stem_b,
                            void* stream);
/* Stem (1x1x1 conv 1 -> 32 channels: mednext stem) + first depthwise 3x3x3 conv in one kernel, the stem output never
 * written: y [N][D][H][W][32] bf16 = dwconv3(stem(x)) with zero padding of the stem output; x [N][D][H][W] fp32;
 * wx [27][32] = w_taps[tap][c]*stem_w[c], wb [27][32] = w_taps[tap][c]*stem_b[c], cst [32] = bias[c] + sum_tap wb[tap][c]
 * (fp32, formed by the caller); stats [N][pytc_stem_dwconv3d_stat_slots][2][32] = per-slot (sum, sum of squares) of the stored
 * values (input of pytc_groupnorm_finalize).
 * W % 16 == 0 runs the matrix-core form (one v_mfma_f32_16x16x32_f16 per 16 voxels x 16 channels over f16 copies of the input
 * and of the fused taps): it reads its A fragments and constants from `mfma_image` (pytc_stem_dwconv3d_mfma_image_bytes bytes,
 * written by pytc_stem_dwconv3d_pack_mfma from the same wx / wb / cst); other widths run the fp32 VALU form (image may be NULL). */
int pytc_stem_dwconv3d_stat_slots(int D, int H, int W);
int pytc_stem_dwconv3d_supported(int C_in, int C, int K);
int pytc_stem_dwconv3d_mfma_image_bytes(void);
int pytc_stem_dwconv3d_pack_mfma(const float* wx, const float* wb, const float* cst, void* image, void* stream);
int pytc_stem_dwconv3d_fwd(const float* x, const float* wx, const float* wb, const float* cst, const void* mfma_image, void* y,
                           float* stats, int N, int D, int H, int W, int C, void* stream);
int pytc_pw_mlp_head_fwd(const pytc_mlp_args* a, const void* head_w, const float* head_b, float* head_y, int n_head,
                         int store_y, void* stream);

/* ---------------------------------------------------------------- dense conv / norm / pool (RSUNet) ---------- */

#define PYTC_ACT_RELU 5
#define PYTC_ACT_LEAKY 6  /* LeakyReLU(slope) and single-weight PReLU */
#define PYTC_ACT_ELU 7

/* Dense Conv3d, stride 1, zero "same" padding, odd kernel per axis, fused pre-activation
 *     y = W * act_in(a*x + b) (+ bias) (+ res)
 * Replaces nn.Conv3d + the preceding NormAct of RSUNet (models/architectures/rsunet.py:87-118, 145-198, 249)
 * and its 1x1 proj / output heads (:249, :388-399).  Weights packed by pytc_conv3d_pack_weight from the
 * PyTorch layout [C_out][C_in][kd][kh][kw]. */
typedef struct {
  const void* x;        /* [N][D][H][W][C_in] */
  const void* w_packed;
  const float* bias;    /* [C_out] or NULL */
  const float* ab;      /* [N][2][C_in] pre-activation affine (norm apply) or NULL */
  const void* res;      /* PYTC_RES_ADD residual [N][D][H][W][C_out] or NULL */
  void* y;
  int N, D, H, W, C_in, C_out;
  int kd, kh, kw;
  int act_in;           /* PYTC_ACT_NONE / RELU / LEAKY / ELU */
  float act_param;      /* negative slope / PReLU weight / ELU alpha */
  int res_mode;         /* PYTC_RES_NONE or PYTC_RES_ADD */
  int dtype;            /* activations and packed weights */
} pytc_conv3d_args;

int64_t pytc_conv3d_packed_elems(int C_out, int C_in, int kd, int kh, int kw, int dtype);
int pytc_conv3d_pack_weight(const float* w, int C_out, int C_in, int kd, int kh, int kw, void* packed, int dtype,
                            void* stream);
int pytc_conv3d_fwd(const pytc_conv3d_args* a, void* stream);

/* Per-(n,c) partial sum / sum of squares of x [N][rows][C] -> stats [N][slots][2][C]
 * (slots = pytc_channel_stats_slots(rows)); the statistics half of GroupNorm / InstanceNorm3d in NormAct. */
int pytc_channel_stats_slots(int64_t rows);
int pytc_channel_stats(const void* x, float* stats, int N, int64_t rows, int C, int dtype, void* stream);
/* GroupNorm with `groups` channel groups: ab[N][2][C] = (gamma*rstd_g, beta - mean_g*gamma*rstd_g). */
int pytc_norm_finalize_groups(const float* stats, int slots, float count, const float* gamma, const float* beta,
                              float eps, int groups, float* ab, int N, int C, void* stream);
/* y = act(a[n][c]*x + b[n][c]) elementwise on [N][rows][C] (ab may be NULL); NormAct outside a conv prologue. */
int pytc_affine_act(const void* x, void* y, const float* ab, int N, int64_t rows, int C, int act, float prm,
                    int dtype, void* stream);
/* nn.MaxPool3d(kernel = stride = (fz,fy,fx)) (rsunet.py:216). */
int pytc_maxpool3d_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int fz, int fy, int fx, int dtype,
                       void* stream);
/* Depthwise ConvTranspose3d with arbitrary per-axis kernel / stride / padding (host int32[3] each); weights fp32
 * [kd*kh*kw][C].  Replaces BilinearUp3d (rsunet.py:33-70). */
int pytc_dwconvT3d_generic_fwd(const void* x, void* y, const float* w, int N, int D, int H, int W, int C,
                               const int32_t* kernel, const int32_t* stride, const int32_t* pad, int dtype,
                               void* stream);

/* ---------------------------------------------------------------- backward (training step) ------------------- */
/* Replaces the autograd backward of the ops above inside ConnectomicsModule.training_step
 * (training/lightning/model.py:863-910 -> Lightning backward).  Two-stage reductions in fixed order
 * (bit-reproducible); gradients of parameters are fp32 in the packed layouts of the forward kernels. */

/* as pytc_groupnorm_finalize, additionally saving mean_rstd [N][2][C] for the backward pass */
int pytc_groupnorm_finalize_mr(const float* stats, int slots, float count, const float* gamma, const float* beta,
                               float eps, float* ab, float* mean_rstd, int N, int C, void* stream);
/* dy == NULL: out = gelu(x);  else out = dy * gelu'(x)   (erf GELU) */
int pytc_gelu(const void* x, const void* dy, void* out, int64_t n, int dtype, void* stream);
int pytc_add_inplace(void* y, const void* x, int64_t n, int dtype, void* stream);
/* dW[o][k] = sum_r dY[r][o] * f(X[r][k]) (f = optional norm affine, as in the forward prologue), db[o] = sum_r dY;
 * workspace: pytc_pw_wgrad_slots(N*rows) * (C_out*C_in + C_out) floats */
int pytc_pw_wgrad_slots(int64_t rows_total);
int pytc_pw_wgrad(const void* x, const float* ab, const void* dy, float* dW, float* db, float* workspace, int N,
                  int64_t rows_per_sample, int C_in, int C_out, int dtype, int x_act /* PYTC_ACT_NONE | PYTC_ACT_GELU:
                  f(X) = gelu(X), the forward's fused pre-activation */, void* stream);
/* dW[tap][c] = sum_{n,o} G[n][o][c] * X[n][o*stride - K/2 + tap][c], db[c] = sum G.  Depthwise conv: G = dL/dy
 * (output grid gdims), X = layer input (xdims).  Transposed depthwise conv (stride 2): G = layer input, X = dL/dy.
 * workspace: pytc_dw_wgrad_slots(...) * (K^3*C + C) floats */
int pytc_dw_wgrad_slots(int N, const int32_t* gdims, const int32_t* xdims, int C, int K, int stride, int dtype);
int pytc_dw_wgrad(const void* g, const void* x, float* dW, float* db, float* workspace, int N, const int32_t* gdims,
                  const int32_t* xdims, int C, int K, int stride, int dtype, void* stream);

/* Deferred-reduction forms (training backward of one block): everything pytc_pw_wgrad / pytc_dw_wgrad do EXCEPT the final
 * slot reduction.  *slots_out = number of partial slots written: dW partials at workspace[0 .. slots*nW), bias partials
 * (want_db != 0) at workspace[slots*nW .. slots*(nW+nB)), nW = C_out*C_in (pw) or K^3*C (dw), nB = C_out or C.  The caller
 * collects the (partials, output, n, slots) items of several gradients and reduces them in ONE launch with
 * pytc_reduce_slots_multi (<= 12 items; per element the same summation tree as the immediate forms, so results are
 * bit-identical).  Replaces the per-gradient reduction launches of the reference's autograd accumulation for a block. */
int pytc_pw_wgrad_partial(const void* x, const float* ab, const void* dy, float* workspace, int want_db, int N,
                          int64_t rows_per_sample, int C_in, int C_out, int dtype, int x_act, int* slots_out, void* stream);
/* pytc_pw_wgrad_partial (x_act = GELU, no affine) fused with the data gradient behind the activation (round 5): one pass over the
 * hidden pre-activation x [N][rows][C_in] and the output gradient dy [N][rows][C_out] gives the weight-gradient partials of the
 * projecting conv (dW[o][k] = sum_r dy[r][o] gelu(x[r][k]), bias partials when want_db) AND dx[r][k] = bf16((sum_o W[o][k] dy[r][o]) *
 * gelu'(x[r][k])), w_t_paired = pytc_pw_pack_weight_paired image of W^T ([C_in][C_out]).  Weight-gradient partials bit-identical to
 * pytc_pw_wgrad_partial; dx equal to pytc_pw_conv_fwd(that paired image, w_paired = 1, PYTC_RES_GELU_BWD) up to one bf16 ulp in a few outputs
 * per million (the GELU' expression is compiled into two kernels); bf16, C_out = 32, C_in in {32, 64}; C_in = 128 (round 6, the
 * 64 -> 128 -> 32 up block: pw_wgrad_dgrad_wide_kernel) multiplies by the derivative of the sigmoid-form GELU the training forward
 * evaluated instead of the erf form's (<= 1.1e-4 apart), weight-gradient partials bit-identical all the same.  Replaces, in the
 * backward of MedNeXtBlock.forward (external nnunet_mednext; contract at mednext_models.py:99-126), autograd's conv3 weight-gradient
 * and conv3 -> act data-gradient nodes. */
int pytc_pw_wgrad_dgrad_supported(int C_in, int C_out, int dtype);
int pytc_pw_wgrad_dgrad_partial(const void* x, const void* dy, const void* w_t_paired, void* dx, float* workspace, int want_db, int N,
                                int64_t rows_per_sample, int C_in, int C_out, int dtype, int* slots_out, void* stream);
/* Backward of a full-resolution block's mixer with the hidden pre-activation REBUILT (round 6): from the depthwise output t
 * [N][rows][32], the forward's GroupNorm affine ab [N][2][32], the output gradient dy [N][rows][32] and the block's weights (w2_paired /
 * w3t_paired = pytc_pw_pack_weight_paired images of W2 (32 -> C_hid) and of W3^T; b2 [C_hid]) one pass gives
 *   dhp [N][rows][C_hid] = bf16((W3^T dy) * gelu'(hp)),  hp = bf16(W2 bf16(a t + b) + b2) recomputed with the forward's arithmetic,
 *   the weight-gradient partials of the projecting conv (dW3[o][k] = sum_r dy[r][o] gelu(hp[r][k]); db3 when want_db3), and, gn != 0,
 *   what pytc_pw_wgrad_groupnorm returns for (t, dhp): s_out [N][2][32], coef [N][3][32], and the per-sample terms of dW2 / db2.
 * workspace: pytc_mixer_bwd_rc_ws_elems(N, rows, C_hid, gn) floats, S = *slots_out = N * pytc_mixer_bwd_rc_sps(...) slots:
 *   dW3 partials [S][32][C_hid] | db3 partials [S][32] | gn: M partials [S][C_hid][32] | q partials [S][C_hid] | term [N][C_hid][32] |
 *   q [N][C_hid];  dW2 = sum_n term[n], db2 = sum_n q[n] and the dW3 / db3 slot sums are left to the caller's pytc_reduce_slots_multi.
 * Same bits as pytc_pw_wgrad_dgrad_partial / pytc_pw_wgrad_groupnorm on the stored hidden tensor whenever those launches' slots do
 * not straddle samples (they group rows by slot; this one never lets a slot straddle).  pytc_mixer_bwd_rc_supported: 0 = no,
 * 1 = gn == 0 only, 2 = both (bf16, C = C_out = 32, C_hid in {32, 64, 96}).  Replaces, in the backward of MedNeXtBlock.forward
 * (external nnunet_mednext; contract at mednext_models.py:99-126), autograd's saved hidden activation, the conv3 weight- and
 * data-gradient nodes, the activation backward and the conv2 weight-gradient node. */
int pytc_mixer_bwd_rc_supported(int C, int C_hid, int C_out, int dtype);
int pytc_mixer_bwd_rc_sps(int N, int64_t rows_per_sample, int C_hid);
int64_t pytc_mixer_bwd_rc_ws_elems(int N, int64_t rows_per_sample, int C_hid, int gn);
int pytc_mixer_bwd_rc(const void* t, const float* ab, const float* mean_rstd, const void* dy, const void* w2_paired, const float* b2,
                      const void* w3t_paired, const float* W2, const float* gamma, float count, void* dhp, float* workspace,
                      float* s_out, float* coef, int want_db3, int gn, int N, int64_t rows_per_sample, int C, int C_hid, int C_out,
                      int dtype, int* slots_out, void* stream);
int pytc_dw_wgrad_partial(const void* g, const void* x, float* workspace, int want_db, int N, const int32_t* gdims,
                          const int32_t* xdims, int C, int K, int stride, int dtype, int* slots_out, void* stream);
/* dst = src (N, D, H, W, C channels-last, dims = {D, H, W}) with the front faces (z, y or x == 0) zeroed: the output gradient of an up
 * block's mixer (training/lightning/model.py:863-910 reaches it through autograd of the external block's F.pad(x, (1,0,1,0,1,0))). */
int pytc_copy_zero_front(const void* src, void* dst, int N, const int32_t* dims, int C, int dtype, void* stream);
typedef struct {
  const float* part;   /* [slots][n] fp32 partials */
  float* out;          /* [n] */
  int64_t n;
  int32_t slots;
  int32_t out_t;       /* 0: out[i]; > 0: the n sums are a [n / out_t][out_t] matrix and `out` receives its transpose (a depthwise
                          weight gradient leaves tap-major [K^3][C] and the parameter is [C][K^3]); fills the struct's former padding */
} pytc_reduce_item;
int pytc_reduce_slots_multi(const pytc_reduce_item* items, int n_items, void* stream);
/* GroupNorm(C,C) backward: dt = rstd*gamma*(dtn - mean_v(dtn) - xhat*mean_v(dtn*xhat)); s_out [N][2][C] holds
 * (sum dtn, sum dtn*xhat) whose sums over N are dbeta / dgamma.  stats_ws: pytc_norm_bwd_ws_elems floats */
int pytc_norm_bwd_ws_elems(int N, int64_t rows, int C);
int pytc_norm_bwd(const void* dtn, const void* t, const float* mean_rstd, const float* gamma, float* stats_ws,
                  float* s_out, void* dt, int N, int64_t rows, float count /* voxels in the statistics */, int C,
                  int dtype, void* stream);
/* GroupNorm-fed expand conv (MedNeXt block: hp = W2 (gamma * xhat + beta) + b2): the weight-gradient sums AND the GroupNorm
 * backward from ONE pass over (t, dhp) plus an epilogue.  With M[n][h][c] = sum_r dhp[r][h] xhat[r][c] and q[n][h] = sum_r dhp[r][h]
 * (the MFMA weight-gradient kernel against xhat in bf16 high + low parts, `sps` row slots per sample),
 *   dW2 [C_hid][C] = sum_n term[n],  term[n] = gamma*M[n] + beta*q[n];   db2 [C_hid] = sum_n q[n]
 *   s_out [N][2][C] = (sum_r dtn, sum_r dtn*xhat) = (sum_h w2[h][c] q[n][h], sum_h w2[h][c] M[n][h][c]),  dtn = w2^T dhp,
 *   w2 = bf16(W2), the data-gradient GEMM's weights;  coef [N][3][C] = (A, B, C) with  dt = A*dtn + B*t + C  the GroupNorm backward
 * -- feed coef to pytc_pw_conv_fwd(res_mode = PYTC_RES_NORM_BWD, res = t, res_bias = coef) of the GEMM that computes dtn: neither
 * the statistics pass nor the apply pass of pytc_norm_bwd runs, and dtn is never stored.  (The statistics alone must NOT be
 * combined with an apply pass over the stored bf16(dtn): they describe the unrounded values.)  mean_rstd / ab: [N][2][C] of
 * pytc_norm_finalize_groups_mr, W2: fp32 [C_hid][C], count = voxels in the statistics.  bf16, C and C_hid multiples of 16
 * (pytc_pw_wgrad_groupnorm_supported).  workspace (pytc_pw_wgrad_groupnorm_ws_elems floats), sps = pytc_pw_wgrad_groupnorm_sps(...):
 *   [N*sps][C_hid*C] dW partials | [N*sps][C_hid] bias partials | term [N][C_hid*C] | q [N][C_hid]
 * the caller reduces term and q over their N slots (pytc_reduce_slots_multi); with sps == 1 the samples' terms are left in the dW partials
 * region instead (term unused: no reduction launch).  Replaces autograd's separate Conv3d-weight and
 * GroupNorm backward passes (reference: torch.nn.GroupNorm + Conv3d under connectomics/training/lightning/model.py:863-910).
 * pytc_norm_bwd_apply: the apply pass of pytc_norm_bwd with s given as [s_parts][N][2][C] (summed over the parts); crop_grid
 * (nullable int32[3], the (D,H,W) grid of the rows): rows on the front faces are dropped and dt is the compact (D-1,H-1,W-1)
 * grid (the zero-padded faces of an up block, MedNeXtUpBlock's F.pad). */
int pytc_pw_wgrad_groupnorm_supported(int C, int C_hid, int dtype);
int pytc_pw_wgrad_groupnorm_sps(int N, int64_t rows_per_sample, int C, int C_hid);
int64_t pytc_pw_wgrad_groupnorm_ws_elems(int N, int64_t rows_per_sample, int C, int C_hid);
int pytc_pw_wgrad_groupnorm(const void* t, const float* mean_rstd, const float* ab, const void* dhp, const float* W2,
                            const float* gamma, float count, float* s_out, float* coef, float* workspace, int N,
                            int64_t rows_per_sample, int C, int C_hid, int dtype, void* stream);
int pytc_norm_bwd_apply(const void* dtn, const void* t, const float* mean_rstd, const float* gamma, const float* s_in,
                        int s_parts, void* dt, int N, int64_t rows, int C, float count, int dtype, const int32_t* crop_grid,
                        void* stream);
/* depthwise conv backward-data, any stride: dx[i] = sum_k dy[(i + K/2 - k)/stride] * w[k] (w: forward taps [K^3][C]) */
int pytc_dwconv3d_bwd_data(const void* dy, const float* w, void* dx, int N, const int32_t* xdims,
                           const int32_t* ydims, int C, int K, int stride, int dtype, void* stream);
/* the same with dx = conv^T(dy) + addend (addend shaped like dx, fp32 sum, one rounding): the gradient of a U-Net skip connection
 * joins the down block's data gradient in this launch (autograd's accumulation of the two is a separate three-pass add). */
int pytc_dwconv3d_bwd_data_add(const void* dy, const float* w, const void* addend, void* dx, int N, const int32_t* xdims,
                               const int32_t* ydims, int C, int K, int stride, int dtype, void* stream);

/* ---------------------------------------------------------------- dense-conv (RSUNet) training ---------- */
/* Backward of the RSUNet building blocks (rsunet.py:73-259 through torch autograd in the reference).  The data
 * gradient of a dense conv is pytc_conv3d_fwd with flipped / transposed weights.
 * pytc_conv3d_wgrad: dW[tap][o][k] = sum_r dY[r][o] * A[r + shift(tap)][k] (A = the conv's activated input, zero padded,
 *   odd kernel sizes, stride 1); workspace: pytc_conv3d_wgrad_ws_elems(...) floats.  bf16 with kh = kw = 3 and channel
 *   counts that are multiples of 16 runs on MFMA (LDS transpose reads), everything else on a VALU kernel.
 * pytc_act_bwd: dt = da * act'(t), t = a[n][c]*x + b[n][c] (ab may be NULL: t = x); dp (may be NULL) = da * min(t, 0),
 *   the summand of the PReLU weight gradient.  act in {RELU, LEAKY (prm = slope), ELU (prm = alpha), NONE}.
 * pytc_norm_finalize_groups_mr: pytc_norm_finalize_groups that also returns (mean, rstd) per (n, c) in mr [N][2][C].
 * pytc_norm_bwd_stats: s_out [N][2][C] = (sum d, sum d * xhat); workspace pytc_norm_bwd_ws_elems floats.
 * pytc_norm_bwd_apply_general: dx = rstd * (gamma*d - M1 - xhat*M2) with M [N][2][C] the means of (gamma*d) and
 *   (gamma*d*xhat) over the statistics group of (n, c), expanded per channel (GroupNorm / InstanceNorm / BatchNorm).
 * pytc_maxpool3d_bwd: dx = 0 except the first maximum of every window, which receives dy (nn.MaxPool3d backward).
 * pytc_dwconv3d_generic_fwd: anisotropic depthwise conv (kernel / stride / pad per axis), the backward-data of
 *   pytc_dwconvT3d_generic_fwd (BilinearUp3d, rsunet.py:33-70). */
 /* pytc_conv3d_pack_weight_dgrad: the packed image of the DATA-GRADIENT conv (C_out -> C_in channels, mirrored taps) read
 *   straight from the forward weight [C_out][C_in][kd][kh][kw]; feed it to pytc_conv3d_fwd with C_in/C_out swapped.
 * pytc_norm_bwd_means: from s [N][2][C] (pytc_norm_bwd_stats): dbeta / dgamma (may be NULL) and the group means M [N][2][C]
 *   of gamma*s that pytc_norm_bwd_apply_general consumes; groups > 0: channel groups within a sample (GroupNorm; groups = C:
 *   InstanceNorm), groups = 0: over the batch (BatchNorm); rows = voxels per sample.
 * pytc_bn_update_running: nn.BatchNorm3d's running_mean / running_var update from the batch (mean, rstd) [2][C]
 *   (unbiased variance, momentum blend); count = N * voxels. */
int pytc_conv3d_pack_weight_dgrad(const float* w, int C_out, int C_in, int kd, int kh, int kw, void* packed, int dtype,
                                  void* stream);
/* Every conv-weight image of a model in ONE launch (the per-step repack of a training run).  pytc_conv3d_pack_plan: the layout
 * pytc_conv3d_pack_weight (direct = 0) / pytc_conv3d_pack_weight_direct (direct = 1) writes for these channel counts: out[0] = kind
 * (0 tap-major, 1 flat chunked), out[1] = KG | KC, out[2] = nchunks, out[3] = G, out[4] = elements written.  pytc_conv3d_pack_multi:
 * table_dev = n_items x 16 int64 on the device { w, packed, s_o, s_c, first block, elements, C_out, C_in, ntap, kind, fp32 image,
 * flip, out[1], out[2], out[3], 0 } with element (o, c, tap) of the conv read from w[o*s_o + c*s_c + (flip ? ntap-1-tap : tap)];
 * total_blocks = sum of ceil(elements / 256).  Images are bit-identical to the single-weight packs. */
int pytc_conv3d_pack_plan(int C_out, int C_in, int kd, int kh, int kw, int dtype, int direct, int64_t* out);
int pytc_conv3d_pack_multi(const int64_t* table_dev, int n_items, int64_t total_blocks, void* stream);
int pytc_norm_bwd_means(const float* s, const float* gamma, float* M, float* dgamma, float* dbeta, int N, int C, int groups,
                        float rows, void* stream);
/* The two group-statistics kernels with an explicit group width (round 4): `groups` groups of `cpg` channels, and the channels from
 * groups * cpg up to C are ALIGNMENT PADDING -- all-zero tails the model adds so that rows are 16-byte aligned when the reference's
 * channel counts are not (RSUNet's stock widths [18, 36, ...], config/profiles/arch_profiles.yaml:34-44, run as 24 / 40 channels):
 * GroupNorm(g, C_real) of rsunet.py:88-94 on the real channels, affine (0, 0) / mean = rstd = 0 / M = 0 on the padding; gamma and
 * beta hold groups * cpg entries.  cpg = 0 is the plain C / groups form; mr may be NULL in the finalize. */
/* pytc_bn_train_finalize with C_real <= C: the channels from C_real on are alignment padding (affine / mean / rstd 0; gamma, beta and
 * the running buffers hold C_real entries) */
int pytc_bn_train_finalize_cpad(const float* stats, int slots_total, float count, const float* gamma, const float* beta, float eps,
                                float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* ab,
                                float* mean_rstd, int N, int C, int C_real, void* stream);
int pytc_norm_finalize_groups_cpg(const float* stats, int slots, float count, const float* gamma, const float* beta, float eps,
                                  int groups, int cpg, float* ab, float* mr, int N, int C, void* stream);
int pytc_norm_bwd_means_cpg(const float* s, const float* gamma, float* M, float* dgamma, float* dbeta, int N, int C, int groups,
                            int cpg, float rows, void* stream);
/* nn.BatchNorm3d in train() mode after the statistics pass, one launch: stats [slots_total][2][C] = the (sum, sum of squares)
 * partials of ALL samples (pytc_channel_stats output viewed flat), count = N * voxels; writes the batch affine ab [N][2][C] and
 * (mean, rstd) [N][2][C] (identical for every n), blends running_mean / running_var (unbiased variance, `momentum`; NULL pair =
 * track_running_stats off) and increments num_batches_tracked (int64, NULL to skip).  Same arithmetic as
 * pytc_norm_finalize_groups_mr(groups = C) + pytc_bn_update_running.  Replaces the BatchNorm half of RSUNet's NormAct
 * (rsunet.py:87-118) and of MONAI's ADN in training. */
int pytc_bn_train_finalize(const float* stats, int slots_total, float count, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* ab,
                           float* mean_rstd, int N, int C, void* stream);
int pytc_bn_update_running(const float* mean_rstd, float* running_mean, float* running_var, int C, float count, float eps,
                           float momentum, void* stream);
int64_t pytc_conv3d_wgrad_ws_elems(int N, int D, int H, int W, int C_in, int C_out, const int32_t* kernel, int dtype);
int pytc_conv3d_wgrad(const void* a, const void* dy, float* dW, float* workspace, int N, int D, int H, int W, int C_in,
                      int C_out, const int32_t* kernel, int dtype, void* stream);
int pytc_act_bwd(const void* da, const void* x, const float* ab, void* dt, void* dp, int N, int64_t rows, int C, int act,
                 float prm, int dtype, void* stream);
int pytc_norm_finalize_groups_mr(const float* stats, int slots, float count, const float* gamma, const float* beta,
                                 float eps, int groups, float* ab, float* mr, int N, int C, void* stream);
int pytc_norm_bwd_stats(const void* dtn, const void* t, const float* mean_rstd, float* stats_ws, float* s_out, int N,
                        int64_t rows, int C, int dtype, void* stream);
int pytc_norm_bwd_apply_general(const void* d, const void* x, const float* mean_rstd, const float* gamma, const float* M,
                                void* dx, int N, int64_t rows, int C, int dtype, void* stream);
/* act_bwd + the norm backward WITHOUT the intermediate dt tensor (RSUNet / MONAI-style conv -> norm -> activation units;
 * reference: torch autograd through nn.BatchNorm3d / GroupNorm + ReLU / PReLU / ELU, rsunet.py + monai BasicUNet blocks): dt = da *
 * act'(a*x + b) is recomputed in registers by the statistics pass and by the apply pass, rounded to the storage type where the
 * three-pass form stored it (same dt values; sums differ by fp32 summation order only).  pytc_act_norm_bwd_stats: s_out [N][2][C] = (sum dt, sum dt * xhat);
 * p_out (nullable, with p_ws) [N][C] = sum da * min(a*x + b, 0), the PReLU weight gradient's summand.  stats_ws:
 * pytc_norm_bwd_ws_elems floats, p_ws half of that.  pytc_act_norm_bwd_apply: dx = rstd * (gamma * dt - M1 - xhat * M2), M from
 * pytc_norm_bwd_means.  C a multiple of 8 (bf16) / 4 (fp32). */
int pytc_act_norm_bwd_stats(const void* da, const void* x, const float* ab, const float* mean_rstd, float* stats_ws, float* s_out,
                            float* p_ws, float* p_out, int N, int64_t rows, int C, int act, float prm, int dtype, void* stream);
int pytc_act_norm_bwd_apply(const void* da, const void* x, const float* ab, const float* mean_rstd, const float* gamma,
                            const float* M, void* dx, int N, int64_t rows, int C, int act, float prm, int dtype, void* stream);
/* the same with gamma holding C_gamma <= C entries (the norm's own channels next to channel-padded activations: no padded copy of gamma) */
int pytc_act_norm_bwd_apply_cg(const void* da, const void* x, const float* ab, const float* mean_rstd, const float* gamma, int C_gamma,
                               const float* M, void* dx, int N, int64_t rows, int C, int act, float prm, int dtype, void* stream);
int pytc_maxpool3d_bwd(const void* x, const void* dy, void* dx, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                       int dtype, void* stream);
int pytc_dwconv3d_generic_fwd(const void* x, void* y, const float* w, int N, int D, int H, int W, int C,
                              const int32_t* kernel, const int32_t* stride, const int32_t* pad, const int32_t* out_dims,
                              int dtype, void* stream);

/* ---------------------------------------------------------------- strided / transposed dense conv (MONAI-style U-Net) - */
/* Replaces the resampling convolutions of the reference's `monai_unet` architecture
 * (connectomics/models/architectures/monai_models.py:197-250 -> monai ResidualUnit / Convolution: nn.Conv3d with stride 2,
 * nn.ConvTranspose3d(k 3, stride 2, padding 1, output_padding 1)) and their torch-autograd backward.
 * pytc_conv3d_strided_fwd: `a` as for pytc_conv3d_fwd but a->D/H/W are the OUTPUT grid and x lives on `in_dims`;
 *   transposed = 0: y[o] = sum_t W[t] * f(x)[o*stride + t - pad];  transposed = 1 (gather form of ConvTranspose3d):
 *   y[o] = sum_t W[t] * f(x)[(o + pad - t) / stride] over the taps where the division is exact.  Weights come from
 *   pytc_conv3d_pack_weight_direct, which reads element (o, c, tap) of the conv the image is FOR at
 *   w[o*s_o + c*s_c + (flip ? ntap-1-tap : tap)] -- nn.Conv3d weight: s_o = C_in*ntap, s_c = ntap; nn.ConvTranspose3d weight
 *   [C_in][C_out][k] used as the forward of the transposed conv or as the data gradient of a strided conv: s_o = ntap,
 *   s_c = C_out*ntap.  The data gradient of a strided conv is the transposed gather, that of a transposed conv the strided
 *   conv, both through this entry.
 * pytc_conv3d_wgrad_strided: dW[o][k][tap] = sum_{r in small grid} small[r][o] * big[r*stride + tap - pad][k]
 *   (strided conv: small = dY, big = conv input; transposed conv: small = conv input, big = dY -> ConvTranspose3d layout);
 *   dW is written [tap][o][k]; workspace pytc_conv3d_wgrad_strided_ws_elems floats; deterministic two-stage sum. */
int64_t pytc_conv3d_direct_packed_elems(int C_out, int C_in, int kd, int kh, int kw, int dtype);
int pytc_conv3d_pack_weight_direct(const float* w, int C_out, int C_in, int kd, int kh, int kw, int64_t s_o, int64_t s_c,
                                   int flip, void* packed, int dtype, void* stream);
/* Round 6 -- the stride-2 transposed gather (ConvTranspose3d k 3 / s 2 / p 1 / output_padding 1, and the data gradient of Conv3d k 3 / s 2 /
 * p 1 on even grids) as EIGHT stride-1 convs on the LDS-tiled kernel of pytc_conv3d_fwd, one per parity (a, b, c) of the output voxel:
 *   y[2i + a, 2j + b, 2k + c] = sum over the (1 + a)(1 + b)(1 + c) sub-taps of W_phase * f(x)[i + dz, j + dy, k + dx]
 * (per axis: even -> tap 1 at offset 0; odd -> tap 2 at offset 0 and tap 0 at offset 1).  bf16, C_in % 8 == 0.
 * pytc_convT3d_phase_plan: out[0..7] element offsets of the eight phase images in ONE buffer, out[8] total elements, out[9] = KC,
 *   out[10] = chunks, out[11..18] = groups per phase; PYTC_ERR_UNSUPPORTED when the shape has no tile plan.
 * The images are written by pytc_conv3d_pack_multi rows of kind 2: { w, image + offset, s_o, s_c, first block, elements, C_out, C_in,
 *   sub-taps of the phase, 2, 0, PHASE (4a + 2b + c), KC, chunks, groups, 0 } with element (o, c, tap) of the 27-tap conv at
 *   w[o*s_o + c*s_c + tap].
 * pytc_convT3d_phase_fwd: `a` as for pytc_conv3d_strided_fwd (a->D/H/W the OUTPUT grid = 2 x in_dims, kernel 3), a->w_packed the images. */
/* pytc_convT3d_c1_fwd: the same transposed conv for ONE output channel (the last up-sampling layer of a single-class U-Net), input-centric:
 * every input voxel forms its 27 tap products once (workspace: 27 * N * Di*Hi*Wi floats, planar), every output voxel sums the 1 .. 8 its
 * parity selects.  w: ConvTranspose3d layout [C_in][1][27] fp32; C_in % 8 == 0. */
int pytc_convT3d_c1_fwd(const void* x, const float* w, const float* bias, void* y, float* workspace, int N, const int32_t* in_dims,
                        int C_in, int dtype, void* stream);
int pytc_convT3d_phase_plan(int C_out, int C_in, int dtype, int64_t* out);
int pytc_convT3d_phase_supported(int C_out, int C_in, int dtype);
int pytc_convT3d_phase_fwd(const pytc_conv3d_args* a, const int32_t* in_dims, void* stream);
int pytc_conv3d_strided_fwd(const pytc_conv3d_args* a, const int32_t* in_dims, const int32_t* stride, const int32_t* pad,
                            int transposed, void* stream);
int64_t pytc_conv3d_wgrad_strided_ws_elems(int N, const int32_t* small_dims, int C_k, int C_o, const int32_t* kernel);
int pytc_conv3d_wgrad_strided(const void* big, const void* small, float* dW, float* workspace, int N,
                              const int32_t* big_dims, const int32_t* small_dims, int C_k, int C_o, const int32_t* kernel,
                              const int32_t* stride, const int32_t* pad, int dtype, void* stream);
/* ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1) forward for C_out <= 4 (the last up-sampling layer of the
 * MONAI-style U-Net, monai_models.py:228-250 with out_channels 1): one thread per output voxel, only the 1..8 taps that reach it;
 * x [N][Di][Hi][Wi][C_in], w fp32 in ConvTranspose3d layout [C_in][C_out][27], y [N][2Di][2Hi][2Wi][C_out] in x's dtype. */
int pytc_convT3d_thin_supported(int C_in, int C_out);
int pytc_convT3d_thin_fwd(const void* x, const float* w, const float* bias, void* y, int N, const int32_t* in_dims, int C_in,
                          int C_out, int dtype, void* stream);

/* ---- train-step epilogue on the device ------------------------------------------------
 * Fused loss  L = w_bce * BCEWithLogits(x, t; weight, pos_weight) + w_dice * Dice(sigmoid(x), t)
 *   (connectomics/models/losses/losses.py:17-44,190-266 WeightedBCEWithLogitsLoss with reduction='mean' - with a weight
 *   map: the mean of weight*bce over the voxels with weight > 0, the map broadcast to the logits' shape - and MONAI DiceLoss(sigmoid=True, smooth_nr, smooth_dr): per (n, c) dice over the spatial
 *   dims, mean over n and c; profiles/loss_profiles.yaml:2-9).  Operands are fp32 (N, C, R) with explicit element strides
 *   {n, c, r} (the network output is channels-last memory viewed as NCDHW); weight may be NULL.
 * pytc_bce_dice_fwd: sums [N*C][5] = (sum_{w>0} w*bce, #{w>0}, sum p*t, sum p, sum t), out [4] = (loss, bce, dice, bce denominator);
 *   workspace pytc_bce_dice_ws_elems floats.  pytc_bce_dice_bwd: dlogits = grad_out[0] * dL/dx from the saved sums.
 * Optimizer (training/optimization/build.py:86-130, trainer.py:321 gradient_clip_val, callbacks.py:869-907 EMA):
 *   table = n_tensors records of 7 int64 {param, grad, exp_avg, exp_avg_sq, ema (0 = none), numel, group}; chunks =
 *   int32 pairs (tensor, chunk index) of pytc_opt_chunk_elems() elements; groups = records of 8 floats {lr, beta1, beta2,
 *   eps, weight_decay, 1-beta1^t, sqrt(1-beta2^t), ema_decay} in HOST memory (at most 8 groups; they travel as kernel
 *   arguments, so the per-step scalars need no H2D copy); table and chunks in device memory.
 * pytc_grad_norm_multi: norm_coef[0] = global L2 norm of the gradients, [1] = min(1, max_norm/(norm+1e-6)) (max_norm <= 0:
 *   1); workspace n_chunks floats.  pytc_adamw_multi: torch.optim.AdamW update of every tensor with the gradient scaled by
 *   norm_coef[1] (NULL: unscaled), then ema = d*ema + (1-d)*param where a record carries an ema pointer.  No host sync. */
/* MedNeXt norm_type='layer' (channels-first LayerNorm, ConvNeXt style): every row [C] of x [rows][C] normalised over its
 * channels, y = gamma * (x - mean) / sqrt(var + eps) + beta (gamma / beta may be NULL); C = VEC * 2^k, 2^k <= 64. */
int pytc_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                        int dtype, void* stream);
/* Backward of the two MedNeXt block variants (reference constructor mednext_models.py:449-463, norm_type='layer' / grn=True;
 * torch autograd through upstream's LayerNorm(channels_first) and GRN branch in the reference's training_step,
 * training/lightning/model.py:863-910).
 * pytc_layernorm_rows_bwd: dx [rows][C] = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = dy * gamma, statistics
 *   recomputed from x; partial [slots][2][C] fp32 (slots = pytc_layernorm_rows_bwd_slots) = per-workgroup sums of dy ([0] ->
 *   dbeta) and dy * xhat ([1] -> dgamma), to be reduced over slots with pytc_reduce_slots / pytc_reduce_slots_multi.
 * pytc_grn_bwd_apply: out = (dh2 * A[n][c] + gelu(hp) * B[n][c]) * gelu'(hp) on [N][rows][C]: the derivative of
 *   h2 = h * (gamma * nx + 1) + beta, h = gelu(hp), with the per-(sample, channel) coefficients A = gamma * nx + 1 and
 *   B = (dL/dgx) / gx built by the caller from the (N, 2, C) sums of pytc_norm_bwd_stats(dh2, h). */
int pytc_layernorm_rows_bwd_slots(int64_t rows, int C, int dtype);
int pytc_layernorm_rows_bwd(const void* dy, const void* x, const float* gamma, void* dx, float* partial, int64_t rows, int C,
                            float eps, int dtype, void* stream);
int pytc_grn_bwd_apply(const void* dh2, const void* hp, const float* A, const float* B, void* out, int N, int64_t rows, int C,
                       int dtype, void* stream);
int64_t pytc_bce_dice_ws_elems(int N, int C, int64_t R);
int pytc_bce_dice_fwd(const float* logits, const float* target, const float* weight, int N, int C, int64_t R,
                      const int64_t* x_strides, const int64_t* t_strides, const int64_t* w_strides, float pos_weight,
                      float w_bce, float w_dice, float smooth_nr, float smooth_dr, float* workspace, float* sums,
                      float* out, void* stream);
int pytc_bce_dice_bwd(const float* logits, const float* target, const float* weight, const float* sums, const float* out,
                      const float* grad_out, float* dlogits, int N, int C, int64_t R, const int64_t* x_strides,
                      const int64_t* t_strides, const int64_t* w_strides, const int64_t* d_strides, float pos_weight,
                      float w_bce, float w_dice, float smooth_nr, float smooth_dr, void* stream);
int pytc_opt_chunk_elems(void);
int pytc_grad_norm_multi(const void* table, const void* chunks, int n_chunks, float max_norm, float* workspace,
                         float* norm_coef, void* stream);
int pytc_adamw_multi(const void* table, const void* chunks, int n_chunks, const float* groups_host, int n_groups,
                     const float* norm_coef, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYTC_HIP_H */
