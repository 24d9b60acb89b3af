// Depthwise transposed conv (K = 3, stride 2, pad 1, bf16 NDHWC, C = 64 / 128), "tile" form: the up blocks' resampling conv writes 8x
// the bytes it reads (zero_() of its output: 210 us at level 0, 6.9 TB/s on MI355X); the cell kernel (dwconv_kernels.hip) took 450 us.
// A workgroup owns ONE tile of 2 x TY x 8 input cells:
//   phase 1: the haloed input tile -> LDS, every 16-byte load issued before the first wait (each input voxel leaves L2 once; the cell
//            kernel read it 8 times through L1);
//   phase 2: a thread keeps the 27 taps of ITS channel pair in registers (the cell kernel: 54 ds_read_b128 of weights per cell and 8
//            channels), walks its share of the tile's cells -- 8 LDS reads, 27 packed FMAs, no 64-bit address arithmetic (32-bit offsets
//            from a wave-uniform base, 24-bit multiplies) -- and never waits for global memory again; the wave's results turn round in
//            a 2 KB LDS tile and leave as two 16-byte stores per lane.
// Arithmetic and its order are the cell kernel's (fp32 FMAs, taps in (a, b, d) order): outputs are BIT-IDENTICAL; statistics leave
// as one partial per workgroup.  Measured (profiles/r04_resample_convs.txt, 8 x 56^3 x 64 -> 112^3): 335-390 us; arithmetic alone
// (statistics-only mode) 180-195 us, stores alone 210 us -- the two ADD here as in every other kernel of this library on this machine,
// so what is left is instruction count: four-byte instead of sixteen-byte stores, four instead of three workgroups per CU, and
// staggered workgroup starts all changed nothing.
#include <mutex>

#include "dwconv_march.h"

namespace pytc {

constexpr int TT_TZ = 2, TT_TX = 8;      // TZ = 2: 31 KB of LDS (C = 64) -> four workgroups per CU; TZ = 4 (52 KB, three) measured slower

template <int C, int TY, bool STORE>
__global__ void __launch_bounds__(256, 4)
dwconvT3d_k3_tile_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, const float* __restrict__ w,
                         const float* __restrict__ bias, float* __restrict__ stats, DwTTile g) {
  constexpr int LPV = C / 2;                           // lanes per voxel (one channel pair each)
  constexpr int SLOTS = 256 / LPV;                     // cells in flight per pass
  constexpr int CELLS = TT_TZ * TY * TT_TX;
  constexpr int EZ = TT_TZ + 1, EY = TY + 1, EX = TT_TX + 1;
  constexpr int NCH = EZ * EY * EX * (C / 8);          // 16-byte chunks of the haloed tile
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned short tile[];     // [EZ][EY][EX][C], then the waves' output tiles
  unsigned short* const otile = tile + EZ * EY * EX * C + (threadIdx.x >> 6) * 1024;   // this wave's 2 KB: [cell slot][8 positions][C]
  __shared__ float red[256 / 64][2][2 * 64];

  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int fx = b % g.tx; b /= g.tx;
  const int fy = b % g.ty;
  const int fz = b / g.ty;
  const int n = blockIdx.y;
  const int z0 = fz * TT_TZ, y0 = fy * TY, x0 = fx * TT_TX;
  const long vcells = (long)g.D * g.H * g.W;
  const unsigned short* xn = x + (long)n * vcells * C;
  unsigned short* yn = y + (long)n * vcells * 8 * C;
  const int Ho = 2 * g.H, Wo = 2 * g.W;

  // ---- phase 1: haloed tile -> LDS (input index clamped into the volume: index -1 feeds the zero faces only, indices past the end
  //      belong to cells that are skipped)
  constexpr int LPT = (NCH + 255) / 256;
  u32x4_t st[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int c = tid + 256 * i;
    const int vox = c / (C / 8), part = c % (C / 8);
    const int lx = vox % EX, t = vox / EX;
    const int ly = t % EY, lz = t / EY;
    const int iz = min(max(z0 - 1 + lz, 0), g.D - 1), iy = min(max(y0 - 1 + ly, 0), g.H - 1), ix = min(max(x0 - 1 + lx, 0), g.W - 1);
    if (c < NCH) st[i] = *reinterpret_cast<const u32x4_t*>(xn + (((long)iz * g.H + iy) * g.W + ix) * C + part * 8);
  }
  // this thread's taps and bias while the tile is on its way
  const int pr = tid % LPV, slot0 = tid / LPV;
  float wt[27][2];
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const float2 v = *reinterpret_cast<const float2*>(w + (long)t * C + pr * 2);
    wt[t][0] = v.x; wt[t][1] = v.y;
  }
  float bv[2] = {0.f, 0.f};
  if (bias) { bv[0] = bias[pr * 2]; bv[1] = bias[pr * 2 + 1]; }
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int c = tid + 256 * i;
    if (c < NCH) *reinterpret_cast<u32x4_t*>(tile + (long)c * 8) = st[i];
  }
  __syncthreads();

  // ---- phase 2.  Address arithmetic is kept out of the cell loop (the first cut spent more issue cycles on 64-bit multiplies and adds
  //      than on the convolution: 250 instructions per cell, 390 us): eight wave-uniform base pointers (one per output parity), one
  //      32-bit byte offset per cell built with 24-bit multiplies (a sample's output is < 4 GB: dwconvT_tile_plan), stores in the
  //      scalar-base + 32-bit-offset form.
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const unsigned int SX = 2u * C * 2u, SY = (unsigned int)Wo * C * 2u * 2u, SZ = (unsigned int)Ho * Wo * C * 2u * 2u;   // bytes per CELL step
  char* const ybase = reinterpret_cast<char*>(yn) + ((((long)2 * z0 * Ho + 2 * y0) * Wo + 2 * x0) * C) * 2;
  // Results leave as 16-byte pieces: a lane computes one channel PAIR (4 bytes) per output, and four-byte stores cost as many
  // memory-pipeline slots as sixteen-byte ones (first cut: 8 dword stores per cell, 190 us of a 370 us launch on top of 180 us of
  // arithmetic -- the two did not overlap); so the wave's 2 KB of results (its cell slots x 8 positions x C channels) turn round in a
  // wave-local LDS tile and go out as two dwordx4 stores per lane.  Piece p = lane + 64 k: voxel p / (C/8) = (cell slot, position q).
  const int lane = tid & 63;
  int pcs[2];                                           // cell slot (within the wave) of this lane's k-th piece
  unsigned int poff[2];                                 // byte offset of that piece relative to its cell's position-0 voxel
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int piece = lane + 64 * k, vox = piece / (C / 8), chunk = piece % (C / 8);
    const int q = vox % 8;
    pcs[k] = vox / 8;
    poff[k] = (unsigned int)(q >> 2) * (SZ / 2) + (unsigned int)((q >> 1) & 1) * (SY / 2) + (unsigned int)(q & 1) * (SX / 2) + chunk * 16u;
  }
  const int wslot0 = (tid >> 6) * (64 / LPV);           // first cell slot of this wave
  f2_t s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
  // tile away from the zero faces and inside the volume: no per-cell masks (workgroup-uniform)
  const bool interior = z0 > 0 && y0 > 0 && x0 > 0 && z0 + TT_TZ <= g.D && y0 + TY <= g.H && x0 + TT_TX <= g.W;
  const f2_t bv2 = {bv[0], bv[1]};
  f2_t wt2[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) wt2[t] = f2_t{wt[t][0], wt[t][1]};
  for (int it = 0; it < CELLS / SLOTS; ++it) {
    const int cell = slot0 + it * SLOTS;
    const int cx = cell % TT_TX, t = cell / TT_TX;
    const int cy = t % TY, cz = t / TY;
    const int mz = z0 + cz, my = y0 + cy, mx = x0 + cx;
    const bool live = interior || (mz < g.D && my < g.H && mx < g.W);
    f2_t xin[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const unsigned int v = *reinterpret_cast<const unsigned int*>(tile + ((((cz + a) * EY + cy + bb) * EX + cx + d) * C + pr * 2));
          xin[a][bb][d] = f2_t{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
        }
    const bool fz0 = !interior && mz == 0, fy0 = !interior && my == 0, fx0 = !interior && mx == 0;
    unsigned int* const orow = reinterpret_cast<unsigned int*>(otile) + ((slot0 - wslot0) * 8) * (C / 2) + pr;
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          f2_t acc = bv2;
          // per axis: even position -> (input 0, tap 2), (input 1, tap 0); odd position -> (input 1, tap 1)
#pragma unroll
          for (int a = pz; a < 2; ++a)
#pragma unroll
            for (int bb = py; bb < 2; ++bb)
#pragma unroll
              for (int d = px; d < 2; ++d) {
                const int kz = pz ? 1 : (a ? 0 : 2), ky = py ? 1 : (bb ? 0 : 2), kx = px ? 1 : (d ? 0 : 2);
                acc = __builtin_elementwise_fma(xin[a][bb][d], wt2[(kz * 3 + ky) * 3 + kx], acc);
              }
          // faces of the padded output grid (position 0 of an axis) are zero; cells beyond the volume contribute nothing
          const bool zero = !live || (pz == 0 && fz0) || (py == 0 && fy0) || (px == 0 && fx0);
          unsigned int bits = __builtin_bit_cast(unsigned int, __builtin_convertvector(acc, bf2_t));
          bits = zero ? 0u : bits;
          if (STORE) orow[(pz * 4 + py * 2 + px) * (C / 2)] = bits;
          const f2_t r = {__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)};
          s1 += r;
          s2 = __builtin_elementwise_fma(r, r, s2);
        }
    if (STORE) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const u32x4_t o = *reinterpret_cast<const u32x4_t*>(otile + (lane + 64 * k) * 8);
        // the cell this piece belongs to: slot wslot0 + pcs[k] of this pass
        const int c2 = wslot0 + pcs[k] + it * SLOTS;
        const int c2x = c2 % TT_TX, t2 = c2 / TT_TX;
        const int c2y = t2 % TY, c2z = t2 / TY;
        const bool ok = interior || (z0 + c2z < g.D && y0 + c2y < g.H && x0 + c2x < g.W);
        const unsigned int off = __umul24((unsigned int)c2z, SZ) + __umul24((unsigned int)c2y, SY) + (unsigned int)c2x * SX + poff[k];
        if (ok) *reinterpret_cast<u32x4_t*>(ybase + off) = o;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (stats) {
    // lanes of one channel pair: pr + LPV * k -- within a wave (LPV = 32: two of them) by shuffle, across waves in LDS, fixed order
    if (LPV == 32) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { s1[i] += __shfl_xor(s1[i], 32, 64); s2[i] += __shfl_xor(s2[i], 32, 64); }
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < LPV) {
      red[wave][0][(lane % LPV) * 2] = s1[0]; red[wave][0][(lane % LPV) * 2 + 1] = s1[1];
      red[wave][1][(lane % LPV) * 2] = s2[0]; red[wave][1][(lane % LPV) * 2 + 1] = s2[1];
    }
    __syncthreads();
    if (tid < 2 * C) {
      const int which = tid / C, ch = tid % C;
      float a = 0.f;
      // a wave covers the pairs (wave * 64 + lane) % LPV: with LPV = 64 every wave holds all of them, with 32 as well
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) a += red[wv][which][ch];
      stats[(((long)n * g.slots + blockIdx.x) * 2 + which) * C + ch] = a;
    }
  }
}

bool dwconvT_tile_plan(DwTTile& g, int N, int D, int H, int W, int C) {
  if (C != 64 && C != 128) return false;
  const int ty = C == 64 ? 8 : 4;
  g.N = N; g.D = D; g.H = H; g.W = W; g.C = C;
  g.tz = (D + TT_TZ - 1) / TT_TZ; g.ty = (H + ty - 1) / ty; g.tx = (W + TT_TX - 1) / TT_TX;
  g.slots = g.tz * g.ty * g.tx;
  // 32-bit byte offsets inside a sample's output, 24-bit multiplies for the cell strides
  return g.slots >= 1 && (long)D * H * W * 8 * C * 2 < (1L << 32) && (long)(2 * H) * (2 * W) * C * 4 < (1L << 24);
}

void dwconvT_tile_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwTTile& g, hipStream_t s) {
  dim3 grid((unsigned)g.slots, (unsigned)g.N), block(256);
  const unsigned short* xp = (const unsigned short*)x;
  unsigned short* yp = (unsigned short*)y;
#define PYTC_TT(CC, TYY, ST)                                                                                                  \
  do {                                                                                                                        \
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&dwconvT3d_k3_tile_kernel<CC, TYY, ST>), 64 * 1024, "dwconvT3d_k3_tile")) return; \
    hipLaunchKernelGGL((dwconvT3d_k3_tile_kernel<CC, TYY, ST>), grid, block,                                                  \
                       (size_t)(TT_TZ + 1) * (TYY + 1) * (TT_TX + 1) * CC * 2 + 4 * 2048, s, xp, yp, w, bias, stats, g);                  \
  } while (0)
  if (g.C == 64) { if (y) PYTC_TT(64, 8, true); else PYTC_TT(64, 8, false); }
  else { if (y) PYTC_TT(128, 4, true); else PYTC_TT(128, 4, false); }
#undef PYTC_TT
}

}  // namespace pytc
