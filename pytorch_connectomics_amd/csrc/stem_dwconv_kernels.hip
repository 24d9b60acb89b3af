// Stem (1x1x1 conv, 1 -> C channels) fused into the first depthwise 3x3x3 conv of the network (MedNeXt level 0).
//
// The stem's output s[v][c] = ws[c]*x[v] + bs[c] is the widest tensor of the network per byte of information (C channels
// derived from ONE input value per voxel); written and re-read it costs 2*C*2 B per voxel.  Here it is never formed:
//     t[v][c] = b1[c] + sum_tap w1[tap][c] * s[v+tap][c]            (s = 0 outside the volume: zero padding of the conv)
//             = (b1[c] + bs[c]*sum_tap w1[tap][c])  +  sum_tap (w1[tap][c]*ws[c]) * x[v+tap]  -  corr(v)[c]
// with corr(v)[c] = bs[c] * sum_{taps outside the volume} w1[tap][c], non-zero only for border voxels (wave-uniform skip
// in the interior).  One lane = 4 consecutive x voxels x 8 channels: the 9 x 6 neighbourhood of the 1-channel input from L1,
// the pre-multiplied taps from LDS once per group, 27*8*4 FMAs, four 16-byte stores (a voxel's 64-byte row by 4 lanes).  The GroupNorm
// statistics of the STORED (bf16-rounded) result are accumulated per lane over its voxels and reduced per workgroup in
// a fixed order through LDS -> per-workgroup slots (summed by pytc_groupnorm_finalize): deterministic, no atomics.
// The un-fused path rounds s to bf16 before the depthwise conv; this one keeps it in fp32 (closer to the fp32 reference).
#include "pytc_common.h"

namespace pytc {

constexpr int SD_C = 32;          // channels (MedNeXt base width)

struct StemDw { int D, H, W, GX; long groups; };   // GX = W / VX groups per x line, groups per sample

// wx [27][C] = w1[tap][c]*ws[c], wb [27][C] = w1[tap][c]*bs[c], cst [C] = b1[c] + sum_tap wb[tap][c] (host-side products),
// copied to LDS once per workgroup.  Lane = (group of 4 consecutive x voxels, 8-channel slice): the 8 x 27 taps of the
// slice are read from LDS once per GROUP (two ds_read_b128 per tap for 4 x 8 FMAs), the 9 x 6 neighbourhood of the
// 1-channel input comes from L1 as one 16-byte and two 4-byte loads per row.  (Earlier forms: one lane = one voxel x 32
// channels with the taps in LDS was LDS-bound at 860 us -- 216 tap reads per voxel; with the taps as scalar operands it
// waited on 54 s_load_dwordx16 per voxel, 781 us.)
// Without the compiler barrier at the top of the group loop hipcc keeps the lane's 216 taps (8 channels x 27) in VGPRs
// across the groups (256 VGPRs, 2 waves/SIMD, no LDS reads in the loop): 985 us against 436 us with the taps re-read from
// LDS per group (122 VGPRs, 4 waves/SIMD).  SD_IT = groups per lane: 2 -> 542 us, 4 -> 436, 8 -> 433, 16 -> 444.
constexpr int SD_IT = 4;
template <int SD_VX>      // consecutive x voxels per lane (4: 406 us at 8x112^3; 8 halves the tap reads but drops to 2 waves/SIMD: 561 us)
__global__ void __launch_bounds__(256)
stem_dwconv_k3_kernel(const float* __restrict__ x, const float* __restrict__ wx, const float* __restrict__ wb,
                      const float* __restrict__ cst, bf16_t* __restrict__ y, float* __restrict__ stats, StemDw g, int slots) {
  __shared__ __attribute__((aligned(16))) float lw[2 * 27 * SD_C + SD_C];   // wx | wb | cst
  __shared__ float red[16][256 + 4];
  for (int i = threadIdx.x; i < 27 * SD_C; i += 256) { lw[i] = wx[i]; lw[27 * SD_C + i] = wb[i]; }
  if (threadIdx.x < SD_C) lw[54 * SD_C + threadIdx.x] = cst[threadIdx.x];
  __syncthreads();
  const float* lwx = lw;
  const float* lwb = lw + 27 * SD_C;
  const float* lc = lw + 54 * SD_C;

  const int n = blockIdx.y;
  const int cg = threadIdx.x & 3, c0 = cg * 8;
  const long rps = (long)g.D * g.H * g.W;
  const float* xn = x + (long)n * rps;
  bf16_t* yn = y + (long)n * rps * SD_C;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  for (int it = 0; it < SD_IT; ++it) {
    asm volatile("" ::: "memory");                  // keep the tap reads of an iteration inside it
    const long G = ((long)blockIdx.x * SD_IT + it) * 64 + (threadIdx.x >> 2);
    const bool live = G < g.groups;
    const long Gc = live ? G : g.groups - 1;
    const int gx = (int)(Gc % g.GX);
    const long tq = Gc / g.GX;
    const int vy = (int)(tq % g.H), vz = (int)(tq / g.H);
    const int x0 = gx * SD_VX;
    float acc[SD_VX][8];
    {
      const float4 k0 = *reinterpret_cast<const float4*>(lc + c0), k1 = *reinterpret_cast<const float4*>(lc + c0 + 4);
#pragma unroll
      for (int v = 0; v < SD_VX; ++v) {
        acc[v][0] = k0.x; acc[v][1] = k0.y; acc[v][2] = k0.z; acc[v][3] = k0.w;
        acc[v][4] = k1.x; acc[v][5] = k1.y; acc[v][6] = k1.z; acc[v][7] = k1.w;
      }
    }
    const bool left_in = x0 > 0, right_in = x0 + SD_VX < g.W;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int dz = r / 3 - 1, dy = r % 3 - 1;
      const int z = vz + dz, yy = vy + dy;
      const bool row_in = z >= 0 && z < g.D && yy >= 0 && yy < g.H;
      const int zc = z < 0 ? 0 : (z >= g.D ? g.D - 1 : z), yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);
      const float* row = xn + ((long)zc * g.H + yc) * g.W + x0;
      const float lft = row[left_in ? -1 : 0], rgt = row[right_in ? SD_VX : SD_VX - 1];
      float in[SD_VX + 2];
      in[0] = (row_in && left_in) ? lft : 0.f;
#pragma unroll
      for (int q = 0; q < SD_VX / 4; ++q) {                               // W % VX == 0: 16-byte aligned
        const float4 mid = *reinterpret_cast<const float4*>(row + q * 4);
        in[1 + q * 4] = row_in ? mid.x : 0.f; in[2 + q * 4] = row_in ? mid.y : 0.f;
        in[3 + q * 4] = row_in ? mid.z : 0.f; in[4 + q * 4] = row_in ? mid.w : 0.f;
      }
      in[SD_VX + 1] = (row_in && right_in) ? rgt : 0.f;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int t = r * 3 + dx;
        const float4 w0 = *reinterpret_cast<const float4*>(lwx + t * SD_C + c0);
        const float4 w1 = *reinterpret_cast<const float4*>(lwx + t * SD_C + c0 + 4);
#pragma unroll
        for (int v = 0; v < SD_VX; ++v) {
          const float xv = in[v + dx];
          acc[v][0] = fmaf(w0.x, xv, acc[v][0]); acc[v][1] = fmaf(w0.y, xv, acc[v][1]);
          acc[v][2] = fmaf(w0.z, xv, acc[v][2]); acc[v][3] = fmaf(w0.w, xv, acc[v][3]);
          acc[v][4] = fmaf(w1.x, xv, acc[v][4]); acc[v][5] = fmaf(w1.y, xv, acc[v][5]);
          acc[v][6] = fmaf(w1.z, xv, acc[v][6]); acc[v][7] = fmaf(w1.w, xv, acc[v][7]);
        }
        // taps that fall outside the volume: the stem output there is 0, not bs -> take bs*w1 back out (borders only)
        const bool edge_l = dx == 0 && !left_in, edge_r = dx == 2 && !right_in;
        if (!row_in || edge_l || edge_r) {
          const float4 b0 = *reinterpret_cast<const float4*>(lwb + t * SD_C + c0);
          const float4 b1 = *reinterpret_cast<const float4*>(lwb + t * SD_C + c0 + 4);
#pragma unroll
          for (int v = 0; v < SD_VX; ++v) {
            const bool out = !row_in || (edge_l && v == 0) || (edge_r && v == SD_VX - 1);
            if (out) {
              acc[v][0] -= b0.x; acc[v][1] -= b0.y; acc[v][2] -= b0.z; acc[v][3] -= b0.w;
              acc[v][4] -= b1.x; acc[v][5] -= b1.y; acc[v][6] -= b1.z; acc[v][7] -= b1.w;
            }
          }
        }
      }
    }
    if (live) {
      const long vbase = ((long)vz * g.H + vy) * g.W + x0;
#pragma unroll
      for (int v = 0; v < SD_VX; ++v) {
        f32x8_t f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = acc[v][j];
        const bf16x8_t o = __builtin_convertvector(f, bf16x8_t);
        *reinterpret_cast<bf16x8_t*>(yn + (vbase + v) * SD_C + c0) = o;
        const f32x8_t q = __builtin_convertvector(o, f32x8_t);            // statistics of what is stored
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += q[j]; s2[j] = fmaf(q[j], q[j], s2[j]); }
      }
    }
  }

  // workgroup reduction in a fixed order: [16 values][256 lanes] scratch, 64 threads sum the 64 lanes of their channel slice
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[j][threadIdx.x] = s1[j]; red[8 + j][threadIdx.x] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < 2 * SD_C) {
    const int which = threadIdx.x / SD_C, c = threadIdx.x % SD_C;
    const float* col = red[which * 8 + (c & 7)];
    float a = 0.f;
    for (int l = 0; l < 64; ++l) a += col[l * 4 + (c >> 3)];
    stats[(((long)n * slots + blockIdx.x) * 2 + which) * SD_C + c] = a;
  }
}

}  // namespace pytc

using namespace pytc;

static int sd_vx(int) { return 4; }

static int sd_it() { return SD_IT; }

extern "C" int pytc_stem_dwconv3d_stat_slots(int D, int H, int W) {
  const long groups = (long)D * H * (W / sd_vx(W));
  const long gpb = 64L * sd_it();       // voxel groups per workgroup (64 groups x 4 channel-group lanes = 256 lanes)
  return (int)((groups + gpb - 1) / gpb);
}

extern "C" int pytc_stem_dwconv3d_supported(int C_in, int C, int K) { return (C_in == 1 && C == SD_C && K == 3) ? 1 : 0; }

extern "C" int pytc_stem_dwconv3d_fwd(const float* x, const float* wx, const float* wb, const float* cst, void* y,
                                      float* stats, int N, int D, int H, int W, int C, void* stream) {
  PYTC_REQUIRE(x && wx && wb && cst && y && stats, "stem_dwconv3d: null pointer");
  PYTC_REQUIRE(N >= 1 && D >= 1 && H >= 1 && W >= 4 && C == SD_C, "stem_dwconv3d: C must be 32");
  PYTC_REQUIRE(W % 4 == 0, "stem_dwconv3d: W must be a multiple of 4");
  const int vx = sd_vx(W);
  StemDw g{D, H, W, W / vx, (long)D * H * (W / vx)};
  const int slots = pytc_stem_dwconv3d_stat_slots(D, H, W);
  hipLaunchKernelGGL(stem_dwconv_k3_kernel<4>, dim3(slots, N), dim3(256), 0, (hipStream_t)stream, x, wx, wb, cst, (bf16_t*)y,
                     stats, g, slots);
  PYTC_LAUNCH_CHECK("stem_dwconv3d");
  return PYTC_OK;
}
