// Depthwise 3x3x3 conv, STRIDE 2, pad 1 (bf16 NDHWC, C = 32 / 64): the down blocks' resampling conv, as a z-march over LDS.
// The gather kernel (dwconv_kernels.hip) fetched 27 taps x 16 bytes per output through L1 with 64-bit address arithmetic and read its
// weights from LDS per tap (54 ds_read_b128 per output and 8 channels): 8 x 112^3 x 32 -> 56^3 took 295-330 us for 0.81 GB (a copy of the
// input alone: 265).  Here a workgroup owns an 8 (x) x TYO (y) footprint of OUTPUT voxels (TYO = 4: 31 KB of LDS at C = 32, four
// workgroups per CU; 8 x 8 footprints measured 5 % slower) and marches along z: the haloed input planes
// (17 x (2 TYO + 1) voxels, all channels) live in a ring of three LDS slots -- two new planes per output plane, requested into registers
// BEFORE the arithmetic of the current plane and written to LDS after it -- so HBM sees the input once (+ 13 % in-plane halo); a thread
// keeps the 27 taps of ITS channel pair in registers (no weight reads at all), reads 27 x 4 bytes of LDS per output, and the plane's
// results leave through a 4 KB LDS tile as 16-byte stores.  fp32 FMAs in (kz, ky, kx) order with zero contributions for taps outside
// the volume = the gather kernel's arithmetic: outputs are BIT-IDENTICAL; statistics leave as one partial per workgroup.
#include <mutex>

#include "dwconv_march.h"

namespace pytc {

template <int C, int TYO>
__global__ void __launch_bounds__(256, 2)
dwconv3d_k3_s2_march_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, const float* __restrict__ w,
                            const float* __restrict__ bias, float* __restrict__ stats, DwS2 g) {
  constexpr int TXO = 8, IY = 2 * TYO + 1, IX = 2 * TXO + 1;
  constexpr int LPV = C / 2, VS = 256 / LPV, OUTV = TYO * TXO, ITEMS = OUTV / VS;
  constexpr int PCH = IY * IX * (C / 8);               // 16-byte chunks per input plane
  constexpr int LPT = (PCH + 255) / 256;               // chunks per thread and plane
  constexpr int PLANE = IY * IX * C;                   // halfwords per ring slot
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  constexpr int OPIECES = OUTV * C / 8;                // 16-byte pieces of the output tile (at most one per thread)
  static_assert(OUTV % VS == 0 && OPIECES <= 256, "at most one 16-byte output piece per thread and plane");
  extern __shared__ __attribute__((aligned(16))) unsigned short ring[];      // [3][IY][IX][C], then the output tile [OUTV][C]
  unsigned short* const otile = ring + 3 * PLANE;
  __shared__ float red[4][2][2 * 64];

  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int fx = b % g.tx; b /= g.tx;
  const int fy = b % g.ty;
  const int zchunk = b / g.ty;
  const int n = blockIdx.y;
  const int yo0 = fy * TYO, xo0 = fx * TXO;
  const int zs = zchunk * g.zc, ze = min(zs + g.zc, g.Do);
  const long in_plane = (long)g.H * g.W * C;
  const unsigned short* xn = x + (long)n * g.D * in_plane;
  unsigned short* yn = y + (long)n * g.Do * g.Ho * g.Wo * C;

  // ---- staging descriptors of this thread's chunks (constant along z)
  int goff[LPT], loff[LPT];
  bool cok[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int c = tid + 256 * i;
    const int vox = c / (C / 8), part = c % (C / 8);
    const int lx = vox % IX, ly = vox / IX;
    const int iy = 2 * yo0 - 1 + ly, ix = 2 * xo0 - 1 + lx;
    cok[i] = c < PCH && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
    goff[i] = (min(max(iy, 0), g.H - 1) * g.W + min(max(ix, 0), g.W - 1)) * C + part * 8;
    loff[i] = c < PCH ? c * 8 : -1;
  }
  auto request = [&](int iz, u32x4_t (&st)[LPT]) {       // every lane loads (clamped); zero-filled when written
    const unsigned short* src = xn + (long)min(max(iz, 0), g.D - 1) * in_plane;
#pragma unroll
    for (int i = 0; i < LPT; ++i) st[i] = *reinterpret_cast<const u32x4_t*>(src + goff[i]);
  };
  auto deposit = [&](int iz, const u32x4_t (&st)[LPT]) { // input plane iz -> ring slot (iz + 1) % 3
    const bool zok = iz >= 0 && iz < g.D;
    unsigned short* dst = ring + ((iz + 1) % 3) * PLANE;
#pragma unroll
    for (int i = 0; i < LPT; ++i)
      if (loff[i] >= 0) *reinterpret_cast<u32x4_t*>(dst + loff[i]) = (zok && cok[i]) ? st[i] : u32x4_t{0u, 0u, 0u, 0u};
  };

  // ---- this thread's channel pair: taps and bias in registers
  const int pr = tid % LPV, vslot = tid / LPV;
  f2_t wt[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const float2 v = *reinterpret_cast<const float2*>(w + (long)t * C + pr * 2);
    wt[t] = f2_t{v.x, v.y};
  }
  f2_t bv = {0.f, 0.f};
  if (bias) bv = f2_t{bias[pr * 2], bias[pr * 2 + 1]};
  // output piece of this thread: voxel tid / (C/8) of the tile (row-major), channels (tid % (C/8)) * 8 ..
  const int ovx = tid / (C / 8), opart = tid % (C / 8);
  const int oyy = yo0 + ovx / TXO, oxx = xo0 + ovx % TXO;
  const bool ook = tid < OPIECES && oyy < g.Ho && oxx < g.Wo;
  const long obase = ((long)oyy * g.Wo + oxx) * C + opart * 8;

  u32x4_t sa[LPT], sb[LPT];
  request(2 * zs - 1, sa);
  request(2 * zs, sb);
  deposit(2 * zs - 1, sa);
  deposit(2 * zs, sb);
  request(2 * zs + 1, sa);
  deposit(2 * zs + 1, sa);
  __syncthreads();

  f2_t s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
  for (int zo = zs; zo < ze; ++zo) {
    const bool more = zo + 1 < ze;
    if (more) { request(2 * zo + 2, sa); request(2 * zo + 3, sb); }    // in flight during this plane's arithmetic
    const unsigned short* p0 = ring + ((2 * zo) % 3) * PLANE;          // input plane 2zo-1 -> slot (2zo) % 3
    const unsigned short* p1 = ring + ((2 * zo + 1) % 3) * PLANE;
    const unsigned short* p2 = ring + ((2 * zo + 2) % 3) * PLANE;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int v = vslot + it * VS;
      const int vy = v / TXO, vx = v % TXO;
      const int base = ((2 * vy) * IX + 2 * vx) * C + pr * 2;
      f2_t acc = bv;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const unsigned short* pl = kz == 0 ? p0 : (kz == 1 ? p1 : p2);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const unsigned int u = *reinterpret_cast<const unsigned int*>(pl + base + (ky * IX + kx) * C);
            acc = __builtin_elementwise_fma(f2_t{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}, wt[(kz * 3 + ky) * 3 + kx], acc);
          }
      }
      const bool live = (yo0 + vy) < g.Ho && (xo0 + vx) < g.Wo;
      unsigned int bits = __builtin_bit_cast(unsigned int, __builtin_convertvector(acc, bf2_t));
      bits = live ? bits : 0u;
      reinterpret_cast<unsigned int*>(otile)[v * (C / 2) + pr] = bits;
      const f2_t r = {__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)};
      s1 += r;
      s2 = __builtin_elementwise_fma(r, r, s2);
    }
    __syncthreads();                                     // the tile is complete; nobody reads the two oldest ring slots any more
    const u32x4_t o = *reinterpret_cast<const u32x4_t*>(otile + (tid < OPIECES ? tid : 0) * 8);
    if (ook) *reinterpret_cast<u32x4_t*>(yn + (long)zo * g.Ho * g.Wo * C + obase) = o;
    if (more) { deposit(2 * zo + 2, sa); deposit(2 * zo + 3, sb); }
    __syncthreads();
  }

  if (stats) {
    // lanes of one channel pair: pr + LPV * k -- inside a wave by shuffle (LPV = 16: four, LPV = 32: two), across waves in LDS, fixed order
#pragma unroll
    for (int off = LPV; off < 64; off <<= 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { s1[i] += __shfl_xor(s1[i], off, 64); s2[i] += __shfl_xor(s2[i], off, 64); }
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < LPV) {
      red[wave][0][lane * 2] = s1[0]; red[wave][0][lane * 2 + 1] = s1[1];
      red[wave][1][lane * 2] = s2[0]; red[wave][1][lane * 2 + 1] = s2[1];
    }
    __syncthreads();
    if (tid < 2 * C) {
      const int which = tid / C, ch = tid % C;
      float a = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) a += red[wv][which][ch];
      stats[(((long)n * g.slots + blockIdx.x) * 2 + which) * C + ch] = a;
    }
  }
}

bool dwconv_s2_plan(DwS2& g, int N, int D, int H, int W, int C) {
  if (C != 32 && C != 64) return false;
  const int tyo = (C == 32 && tuning_get("dwconv_s2_tyo4", 1) == 0) ? 8 : 4;   // 4 x 8 outputs: 31 KB of LDS, four workgroups per CU (8 x 8: 60 KB, two; 5 % slower)
  g.tyo = tyo;
  g.N = N; g.D = D; g.H = H; g.W = W; g.C = C;
  g.Do = (D - 1) / 2 + 1; g.Ho = (H - 1) / 2 + 1; g.Wo = (W - 1) / 2 + 1;
  g.ty = (g.Ho + tyo - 1) / tyo; g.tx = (g.Wo + 7) / 8;
  // z-chunks: >= ~128 workgroups per sample (the split must not depend on N: batch-invariant statistics), chunks of >= 7 output planes
  int nzc = (128 + g.ty * g.tx - 1) / (g.ty * g.tx);
  const int maxc = g.Do / 7 < 1 ? 1 : g.Do / 7;
  if (nzc > maxc) nzc = maxc;
  if (nzc < 1) nzc = 1;
  g.zc = (g.Do + nzc - 1) / nzc;
  g.nzc = (g.Do + g.zc - 1) / g.zc;
  g.slots = g.ty * g.tx * g.nzc;
  return (long)H * W * C < (1L << 30);
}

void dwconv_s2_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwS2& g, hipStream_t s) {
  dim3 grid((unsigned)g.slots, (unsigned)g.N), block(256);
  const unsigned short* xp = (const unsigned short*)x;
  unsigned short* yp = (unsigned short*)y;
#define PYTC_S2(CC, TYY)                                                                                                      \
  do {                                                                                                                        \
    const size_t lds = (size_t)(3 * (2 * TYY + 1) * 17 * CC + TYY * 8 * CC) * 2;                                               \
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&dwconv3d_k3_s2_march_kernel<CC, TYY>), 72 * 1024, "dwconv3d_k3_s2_march")) return; \
    hipLaunchKernelGGL((dwconv3d_k3_s2_march_kernel<CC, TYY>), grid, block, lds, s, xp, yp, w, bias, stats, g);               \
  } while (0)
  if (g.C == 32 && g.tyo == 8) PYTC_S2(32, 8);
  else if (g.C == 32) PYTC_S2(32, 4);
  else PYTC_S2(64, 4);
#undef PYTC_S2
}

}  // namespace pytc
