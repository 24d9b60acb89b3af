// Stem (1x1x1 conv, 1 -> C channels) fused into the first depthwise 3x3x3 conv of the network (MedNeXt level 0).
//
// The stem's output s[v][c] = ws[c]*x[v] + bs[c] is the widest tensor of the network per byte of information (C channels
// derived from ONE input value per voxel); written and re-read it costs 2*C*2 B per voxel.  Here it is never formed:
//     t[v][c] = b1[c] + sum_tap w1[tap][c] * s[v+tap][c]            (s = 0 outside the volume: zero padding of the conv)
//             = (b1[c] + bs[c]*sum_tap w1[tap][c])  +  sum_tap (w1[tap][c]*ws[c]) * x[v+tap]  -  corr(v)[c]
// with corr(v)[c] = bs[c] * sum_{taps outside the volume} w1[tap][c], non-zero only for border voxels (wave-uniform skip
// in the interior).  One lane = one voxel x all C channels: 27 neighbour values of the 1-channel input from L1 (4 B each),
// the pre-multiplied taps as scalar (SGPR) operands, 27*C FMAs, one 64-byte row store.  The GroupNorm
// statistics of the STORED (bf16-rounded) result are accumulated per lane over its voxels and reduced per workgroup in
// a fixed order through LDS -> per-workgroup slots (summed by pytc_groupnorm_finalize): deterministic, no atomics.
// The un-fused path rounds s to bf16 before the depthwise conv; this one keeps it in fp32 (closer to the fp32 reference).
#include "pytc_common.h"

namespace pytc {

constexpr int SD_C = 32;          // channels (MedNeXt base width)
constexpr int SD_VPL = 8;         // voxels per lane
constexpr int SD_VPB = 256 * SD_VPL;

struct StemDw { int D, H, W; long rps; };

// wx [27][C] = w1[tap][c]*ws[c], wb [27][C] = w1[tap][c]*bs[c], cst [C] = b1[c] + sum_tap wb[tap][c] (host-side products).
// They are read with compile-time indices from kernel-argument pointers: wave-uniform, so hipcc turns them into scalar
// loads (s_load_dwordx8) and the FMAs take the tap as an SGPR operand -- no LDS traffic for the 864 taps (an LDS-resident
// copy made the kernel LDS-bound: 216 broadcast ds_read_b128 per voxel against 864 FMAs, 860 us).
__global__ void __launch_bounds__(256, 2)
stem_dwconv_k3_kernel(const float* __restrict__ x, const float* __restrict__ wx, const float* __restrict__ wb,
                      const float* __restrict__ cst, bf16_t* __restrict__ y, float* __restrict__ stats, StemDw g, int slots) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 256];     // statistics scratch of the final reduction
  const int n = blockIdx.y;
  const float* xn = x + (long)n * g.rps;
  bf16_t* yn = y + (long)n * g.rps * SD_C;
  float s1[SD_C], s2[SD_C];
#pragma unroll
  for (int c = 0; c < SD_C; ++c) { s1[c] = 0.f; s2[c] = 0.f; }

  const long v0 = (long)blockIdx.x * SD_VPB;
  for (int it = 0; it < SD_VPL; ++it) {
    const long v = v0 + it * 256 + threadIdx.x;                 // consecutive lanes = consecutive x: coalesced rows
    const bool live = v < g.rps;
    const long vc = live ? v : g.rps - 1;
    const int vx = (int)(vc % g.W);
    const long tq = vc / g.W;
    const int vy = (int)(tq % g.H), vz = (int)(tq / g.H);
    float xn27[27];
    unsigned outside = 0u;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int dz = t / 9 - 1, dy = (t / 3) % 3 - 1, dx = t % 3 - 1;
      const int z = vz + dz, yy = vy + dy, xx = vx + dx;
      const bool in = z >= 0 && z < g.D && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      // clamped address + select: no branch around the load
      const int zc = z < 0 ? 0 : (z >= g.D ? g.D - 1 : z), yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy),
                xc = xx < 0 ? 0 : (xx >= g.W ? g.W - 1 : xx);
      const float val = xn[((long)zc * g.H + yc) * g.W + xc];
      xn27[t] = in ? val : 0.f;
      outside |= in ? 0u : (1u << t);
    }
    float acc[SD_C];
#pragma unroll
    for (int c = 0; c < SD_C; ++c) acc[c] = cst[c];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const float xv = xn27[t];
      // the taps are loop-invariant: without this hipcc hoists all 864 scalar loads out of the voxel loop and spills SGPRs
      if (t % 2 == 0) asm volatile("" ::: "memory");
#pragma unroll
      for (int c = 0; c < SD_C; ++c) acc[c] = fmaf(wx[t * SD_C + c], xv, acc[c]);
    }
    if (__builtin_amdgcn_ballot_w64(outside != 0u) != 0ull) {     // some lane of the wave sits on the border
      for (int t = 0; t < 27; ++t) {
        if (outside & (1u << t)) {
#pragma unroll
          for (int c = 0; c < SD_C; ++c) acc[c] -= wb[t * SD_C + c];
        }
      }
    }
    if (live) {
#pragma unroll
      for (int c8 = 0; c8 < SD_C / 8; ++c8) {
        f32x8_t f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = acc[c8 * 8 + j];
        const bf16x8_t o = __builtin_convertvector(f, bf16x8_t);
        *reinterpret_cast<bf16x8_t*>(yn + v * SD_C + c8 * 8) = o;
        const f32x8_t r = __builtin_convertvector(o, f32x8_t);    // statistics of what is stored
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[c8 * 8 + j] += r[j]; s2[c8 * 8 + j] = fmaf(r[j], r[j], s2[c8 * 8 + j]); }
      }
    }
  }

  // workgroup reduction in a fixed order: lane-major scratch [64 values][256 lanes], then 64 threads sum one value each
#pragma unroll
  for (int c = 0; c < SD_C; ++c) { lds[c * 256 + threadIdx.x] = s1[c]; lds[(SD_C + c) * 256 + threadIdx.x] = s2[c]; }
  __syncthreads();
  if (threadIdx.x < 2 * SD_C) {
    const float* col = lds + threadIdx.x * 256;
    float a = 0.f;
    for (int l = 0; l < 256; ++l) a += col[(l + threadIdx.x) & 255];     // rotated start: conflict-free, fixed per thread
    const int which = threadIdx.x / SD_C, c = threadIdx.x % SD_C;
    stats[(((long)n * slots + blockIdx.x) * 2 + which) * SD_C + c] = a;
  }
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_stem_dwconv3d_stat_slots(int D, int H, int W) {
  const long rps = (long)D * H * W;
  return (int)((rps + SD_VPB - 1) / SD_VPB);
}

extern "C" int pytc_stem_dwconv3d_supported(int C_in, int C, int K) { return (C_in == 1 && C == SD_C && K == 3) ? 1 : 0; }

extern "C" int pytc_stem_dwconv3d_fwd(const float* x, const float* wx, const float* wb, const float* cst, void* y,
                                      float* stats, int N, int D, int H, int W, int C, void* stream) {
  PYTC_REQUIRE(x && wx && wb && cst && y && stats, "stem_dwconv3d: null pointer");
  PYTC_REQUIRE(N >= 1 && D >= 1 && H >= 1 && W >= 1 && C == SD_C, "stem_dwconv3d: C must be 32");
  StemDw g{D, H, W, (long)D * H * W};
  const int slots = pytc_stem_dwconv3d_stat_slots(D, H, W);
  hipLaunchKernelGGL(stem_dwconv_k3_kernel, dim3(slots, N), dim3(256), 0, (hipStream_t)stream, x, wx, wb, cst, (bf16_t*)y,
                     stats, g, slots);
  PYTC_LAUNCH_CHECK("stem_dwconv3d");
  return PYTC_OK;
}
