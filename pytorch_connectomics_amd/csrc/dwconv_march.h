// Geometry of the z-march depthwise kernels (dwconv_kernels.hip: VALU forms; dwconv_mfma_kernels.hip: matrix-core form).
#pragma once
#include "pytc_common.h"

namespace pytc {

struct DwMarch {
  int N, D, H, W, C;
  int ty, tx, zc, nzc;   // footprints per axis, z-chunk length, z-chunks
  int tilex;             // x extent of a footprint (8, or 16 for the 512-thread forward variant)
  int slots;             // workgroups per (sample, channel group)
  int swizzle;           // XCD-aware block remap on/off
  int cg_inner;          // block order: channel group fastest (else slot fastest)
};

// dwconv_mfma_kernels.hip: bf16, K = 3, stride 1, C % 32 == 0, 8 x 8 footprints (g.tilex == 8); grid = slots * (C / 32) * N blocks.
// variant (knob dwconv_mfma_variant): bit 0 = hi + lo weights, bit 1 = two planes in flight.
void dwconv_mfma_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwMarch& g, int variant,
                        hipStream_t s);

// Fused block on the same kernel (see the comment at DwMix's use in dwconv_mfma_kernels.hip)
struct DwMix {
  const bf16x8_t* w2n;     // per-sample paired bf16 images of W2 * diag(a_n): [N][MIXHC * 2 * 64] fragments (pytc_groupnorm_fold_mlp)
  const float* b2n;        // [N][32 * MIXHC] folded expand bias
  const h8_t* w3;          // paired fp16 image of the projecting conv [2][MIXHC][64]
  const float* b3;         // [32]
  long w2_stride;          // fragments per sample image
  int residual;            // 1: y = mixer + x
  // output head in the epilogue (pw_mlp_kernel HEAD): logits[o] = head_b[o] + sum_c head_w[o][c] * bf16(y[c]), o < n_head <= 16
  const bf16x8_t* head_w;
  const float* head_b;
  float* head_y;           // [N][D*H*W][n_head] fp32
  int n_head, store_y;
  float* prof;             // PROBE 4 only (knob dwconv_mfma_probe = 4): [N][slots][4 waves][8] section cycle sums
};
int dwmix_launch(const void* x, void* y, const float* w, const float* bias, const DwMarch& g, const DwMix& mx, int c_hid, int variant,
                 hipStream_t s);

// dwconvT_tile_kernels.hip: transposed K = 3 / stride 2 conv, bf16, C = 64 / 128, one tile of input cells per workgroup
struct DwTTile { int N, D, H, W, C, tz, ty, tx, slots; };
bool dwconvT_tile_plan(DwTTile& g, int N, int D, int H, int W, int C);
void dwconvT_tile_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwTTile& g, hipStream_t s);

// dwconv_s2_kernels.hip: K = 3 / stride 2 conv of the down blocks, bf16, C = 32 / 64, z-march over an LDS ring of input planes
struct DwS2 { int N, D, H, W, C, Do, Ho, Wo, ty, tx, zc, nzc, slots, tyo; };
bool dwconv_s2_plan(DwS2& g, int N, int D, int H, int W, int C);
void dwconv_s2_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwS2& g, hipStream_t s);

}  // namespace pytc
