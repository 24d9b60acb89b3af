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

// dwconvT_tile_kernels.hip: transposed K = 3 / stride 2 conv, bf16, C = 64 / 128, one tile of input cells per workgroup
struct DwTTile { int N, D, H, W, C, tz, ty, tx, slots; };
bool dwconvT_tile_plan(DwTTile& g, int N, int D, int H, int W, int C);
void dwconvT_tile_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwTTile& g, hipStream_t s);

// dwconv_s2_kernels.hip: K = 3 / stride 2 conv of the down blocks, bf16, C = 32 / 64, z-march over an LDS ring of input planes
struct DwS2 { int N, D, H, W, C, Do, Ho, Wo, ty, tx, zc, nzc, slots, tyo; };
bool dwconv_s2_plan(DwS2& g, int N, int D, int H, int W, int C);
void dwconv_s2_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwS2& g, hipStream_t s);

}  // namespace pytc
