// Depthwise 3-D convolution kernels (NDHWC, lanes along C so every global access is a
// coalesced vector of VEC channels), fused with the per-(n,c) sum / sum-of-squares the
// following GroupNorm(C, C) needs.  Statistics are written as per-workgroup partials and
// reduced in a fixed order by groupnorm_finalize -> bit-reproducible, no float atomics.
#include <type_traits>
#include "pytc_common.h"
#include "dwconv_march.h"

namespace pytc {

struct DwGeom {
  int N, D, H, W, C, K, stride, Do, Ho, Wo;
  int lpv;        // lanes per voxel = C / VEC
  int vs;         // voxel slots per pass = 256 / lpv
  int iters;      // passes per workgroup
  int slots;      // workgroups per sample
  int cell;       // transposed K=3: work items are 2x2x2 output cells (one per input voxel)
  int xblock;     // K=5/7 stride 1: work items are groups of 4 consecutive x outputs
};

template <int VEC>
__device__ __forceinline__ void block_stats_reduce(float (&s1)[VEC], float (&s2)[VEC], int cv, int vslot,
                                                   bool active, const DwGeom& g, float* __restrict__ stats,
                                                   int n, int slot, float* lds) {
  // lds: [vs][2][C-of-this-block] ; deterministic serial reduce over vslot
  const int C = g.C;
  if (active) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      lds[(vslot * 2 + 0) * C + cv * VEC + i] = s1[i];
      lds[(vslot * 2 + 1) * C + cv * VEC + i] = s2[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
    int which = c / C, ch = c % C;
    float acc = 0.f;
    for (int v = 0; v < g.vs; ++v) acc += lds[(v * 2 + which) * C + ch];
    stats[(((long)n * g.slots + slot) * 2 + which) * C + ch] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Generic direct kernel: any K (templated for unrolling), stride 1 or 2, any C with C % VEC == 0.
template <typename T, int VEC, int K>
__global__ void __launch_bounds__(256)
dwconv3d_direct_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w,
                       const float* __restrict__ bias, float* __restrict__ stats, DwGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = blockIdx.y, slot = blockIdx.x;
  const int cv = threadIdx.x % g.lpv, vslot = threadIdx.x / g.lpv;
  const bool lane_ok = vslot < g.vs;
  const long vout = (long)g.Do * g.Ho * g.Wo;
  const int pad = K / 2;
  const int C = g.C;
  float s1[VEC], s2[VEC], bv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; bv[i] = bias ? bias[cv * VEC + i] : 0.f; }
  const T* xn = x + (long)n * g.D * g.H * g.W * C;
  T* yn = y + (long)n * vout * C;
  for (int it = 0; it < g.iters; ++it) {
    long v = ((long)slot * g.iters + it) * g.vs + vslot;
    if (!lane_ok || v >= vout) continue;
    int ox = (int)(v % g.Wo);
    long t = v / g.Wo;
    int oy = (int)(t % g.Ho);
    int oz = (int)(t / g.Ho);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = bv[i];
#pragma unroll
    for (int kz = 0; kz < K; ++kz) {
      int iz = oz * g.stride - pad + kz;
      if (iz < 0 || iz >= g.D) continue;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        int iy = oy * g.stride - pad + ky;
        if (iy < 0 || iy >= g.H) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          int ix = ox * g.stride - pad + kx;
          if (ix < 0 || ix >= g.W) continue;
          float xv[VEC], wv[VEC];
          VecIO<T, VEC>::load(xn + (((long)iz * g.H + iy) * g.W + ix) * C + cv * VEC, xv);
          VecIO<float, VEC>::load(w + ((kz * K + ky) * K + kx) * C + cv * VEC, wv);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(xv[i], wv[i], acc[i]);
        }
      }
    }
    VecIO<T, VEC>::store(yn + v * C + cv * VEC, acc);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float r = to_f32<T>(from_f32<T>(acc[i]));  // statistics of the values as stored
      s1[i] += r;
      s2[i] = fmaf(r, r, s2[i]);
    }
  }
  if (stats) block_stats_reduce<VEC>(s1, s2, cv, vslot, lane_ok, g, stats, n, slot, lds);
}

// ---------------------------------------------------------------------------------------------
// K = 3 gather form without branches (stride 1 or 2; the down blocks and the volumes too small for the z-march):
// every tap is loaded from a clamped address (a branch around a load makes hipcc wait for each one separately) and an
// out-of-range tap gets a zero weight instead; the 27 x C taps are staged once per workgroup in LDS.
template <typename T, int VEC, int K>
__global__ void __launch_bounds__(256)
dwconv3d_k3_gather_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w,
                          const float* __restrict__ bias, float* __restrict__ stats, DwGeom g) {
  // K = 3 keeps its 3x3 plane of inputs in registers; K = 5 / 7 go row by row (K vectors live) -- same clamped,
  // branch-free loads and zero weights for the taps that fall outside
  extern __shared__ __attribute__((aligned(16))) float lds[];   // K^3*C taps, later the statistics scratch
  constexpr int P = K / 2;
  const int n = blockIdx.y, slot = blockIdx.x;
  const int cv = threadIdx.x % g.lpv, vslot = threadIdx.x / g.lpv;
  const bool lane_ok = vslot < g.vs;
  const long vout = (long)g.Do * g.Ho * g.Wo;
  const int C = g.C;
  for (int i = threadIdx.x; i < K * K * K * C; i += 256) lds[i] = w[i];
  __syncthreads();
  float s1[VEC], s2[VEC], bv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; bv[i] = bias ? bias[cv * VEC + i] : 0.f; }
  const T* xn = x + (long)n * g.D * g.H * g.W * C + cv * VEC;
  T* yn = y + (long)n * vout * C + cv * VEC;
  const float* wl = lds + cv * VEC;
  for (int it = 0; it < g.iters; ++it) {
    const long v = ((long)slot * g.iters + it) * g.vs + vslot;
    if (!lane_ok || v >= vout) continue;
    const int ox = (int)(v % g.Wo);
    const long t = v / g.Wo;
    const int oy = (int)(t % g.Ho), oz = (int)(t / g.Ho);
    int zi[K], yi[K], xi[K];
    bool zv[K], yv[K], xv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int iz = oz * g.stride - P + k, iy = oy * g.stride - P + k, ix = ox * g.stride - P + k;
      zv[k] = iz >= 0 && iz < g.D; yv[k] = iy >= 0 && iy < g.H; xv[k] = ix >= 0 && ix < g.W;
      zi[k] = min(max(iz, 0), g.D - 1); yi[k] = min(max(iy, 0), g.H - 1); xi[k] = min(max(ix, 0), g.W - 1);
    }
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = bv[i];
    if constexpr (K == 3) {
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        float in[9][VEC];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            VecIO<T, VEC>::load(xn + (((long)zi[kz] * g.H + yi[ky]) * g.W + xi[kx]) * C, in[ky * 3 + kx]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bool ok = zv[kz] & yv[ky] & xv[kx];
            float wv[VEC];
            VecIO<float, VEC>::load(wl + ((kz * 3 + ky) * 3 + kx) * C, wv);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(in[ky * 3 + kx][i], ok ? wv[i] : 0.f, acc[i]);
          }
      }
    } else {
      for (int kz = 0; kz < K; ++kz) {
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const T* row = xn + ((long)zi[kz] * g.H + yi[ky]) * g.W * C;
          float in[K][VEC];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) VecIO<T, VEC>::load(row + (long)xi[kx] * C, in[kx]);
          const bool zy = zv[kz] & yv[ky];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const bool ok = zy & xv[kx];
            float wv[VEC];
            VecIO<float, VEC>::load(wl + ((kz * K + ky) * K + kx) * C, wv);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(in[kx][i], ok ? wv[i] : 0.f, acc[i]);
          }
        }
      }
    }
    VecIO<T, VEC>::store(yn + v * C, acc);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float r = to_f32<T>(from_f32<T>(acc[i]));
      s1[i] += r;
      s2[i] = fmaf(r, r, s2[i]);
    }
  }
  __syncthreads();
  if (stats) block_stats_reduce<VEC>(s1, s2, cv, vslot, lane_ok, g, stats, n, slot, lds);
}

// ---------------------------------------------------------------------------------------------
// Depthwise transposed conv, stride 2, pad K/2 (gather form).  Output grid is (2D,2H,2W): position
// p holds convT output o = p - 1; the p == 0 faces are written as zero and excluded from stats.
template <typename T, int VEC, int K>
__global__ void __launch_bounds__(256)
dwconvT3d_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w,
                 const float* __restrict__ bias, float* __restrict__ stats, DwGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = blockIdx.y, slot = blockIdx.x;
  const int cv = threadIdx.x % g.lpv, vslot = threadIdx.x / g.lpv;
  const bool lane_ok = vslot < g.vs;
  const long vout = (long)g.Do * g.Ho * g.Wo;  // = (2D)(2H)(2W)
  const int pad = K / 2;
  const int C = g.C;
  float s1[VEC], s2[VEC], bv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; bv[i] = bias ? bias[cv * VEC + i] : 0.f; }
  const T* xn = x + (long)n * g.D * g.H * g.W * C;
  T* yn = y + (long)n * vout * C;
  for (int it = 0; it < g.iters; ++it) {
    long v = ((long)slot * g.iters + it) * g.vs + vslot;
    if (!lane_ok || v >= vout) continue;
    int px = (int)(v % g.Wo);
    long t = v / g.Wo;
    int py = (int)(t % g.Ho);
    int pz = (int)(t / g.Ho);
    float acc[VEC];
    if (px == 0 || py == 0 || pz == 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      VecIO<T, VEC>::store(yn + v * C + cv * VEC, acc);
      continue;
    }
    int oz = pz - 1, oy = py - 1, ox = px - 1;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = bv[i];
    if constexpr (K == 3) {
      // stride 2, pad 1: even o -> one tap (k=1, i=o/2); odd o -> two taps (k=0, i=(o+1)/2), (k=2, i=(o-1)/2);
      // o <= 2D-2 keeps every index in range: no bounds checks
      int iz[2], kz[2], iy[2], ky[2], ix[2], kx[2];
      const int nz = 1 + (oz & 1), ny = 1 + (oy & 1), nx = 1 + (ox & 1);
      if (oz & 1) { iz[0] = (oz + 1) >> 1; kz[0] = 0; iz[1] = (oz - 1) >> 1; kz[1] = 2; } else { iz[0] = oz >> 1; kz[0] = 1; iz[1] = 0; kz[1] = 0; }
      if (oy & 1) { iy[0] = (oy + 1) >> 1; ky[0] = 0; iy[1] = (oy - 1) >> 1; ky[1] = 2; } else { iy[0] = oy >> 1; ky[0] = 1; iy[1] = 0; ky[1] = 0; }
      if (ox & 1) { ix[0] = (ox + 1) >> 1; kx[0] = 0; ix[1] = (ox - 1) >> 1; kx[1] = 2; } else { ix[0] = ox >> 1; kx[0] = 1; ix[1] = 0; kx[1] = 0; }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (a < nz) {
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2) {
            if (b2 < ny) {
#pragma unroll
              for (int c2 = 0; c2 < 2; ++c2) {
                if (c2 < nx) {
                  float xv[VEC], wv[VEC];
                  VecIO<T, VEC>::load(xn + (((long)iz[a] * g.H + iy[b2]) * g.W + ix[c2]) * C + cv * VEC, xv);
                  VecIO<float, VEC>::load(w + ((kz[a] * 3 + ky[b2]) * 3 + kx[c2]) * C + cv * VEC, wv);
#pragma unroll
                  for (int i = 0; i < VEC; ++i) acc[i] = fmaf(xv[i], wv[i], acc[i]);
                }
              }
            }
          }
        }
      }
    } else {
#pragma unroll
    for (int kz = 0; kz < K; ++kz) {
      int tz = oz + pad - kz;
      if (tz < 0 || (tz & 1) || (tz >> 1) >= g.D) continue;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        int ty = oy + pad - ky;
        if (ty < 0 || (ty & 1) || (ty >> 1) >= g.H) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          int tx = ox + pad - kx;
          if (tx < 0 || (tx & 1) || (tx >> 1) >= g.W) continue;
          float xv[VEC], wv[VEC];
          VecIO<T, VEC>::load(xn + (((long)(tz >> 1) * g.H + (ty >> 1)) * g.W + (tx >> 1)) * C + cv * VEC, xv);
          VecIO<float, VEC>::load(w + ((kz * K + ky) * K + kx) * C + cv * VEC, wv);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(xv[i], wv[i], acc[i]);
        }
      }
    }
    }
    VecIO<T, VEC>::store(yn + v * C + cv * VEC, acc);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float r = to_f32<T>(from_f32<T>(acc[i]));
      s1[i] += r;
      s2[i] = fmaf(r, r, s2[i]);
    }
  }
  if (stats) block_stats_reduce<VEC>(s1, s2, cv, vslot, lane_ok, g, stats, n, slot, lds);
}


// K = 5 / 7, stride 1: one lane = XB consecutive x outputs of its VEC channels.  A (kz, ky) row of the stencil needs
// XB + K - 1 input vectors for XB outputs (instead of XB * K) and every tap vector is read from LDS once per XB outputs:
// the plain gather form spends its time issuing loads (125 / 343 per output).
template <typename T, int VEC, int K, int XB>
__global__ void __launch_bounds__(256)
dwconv3d_xblock_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w,
                       const float* __restrict__ bias, float* __restrict__ stats, DwGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // K^3*C taps, later the statistics scratch
  constexpr int P = K / 2, NI = XB + K - 1;
  const int n = blockIdx.y, slot = blockIdx.x;
  const int cv = threadIdx.x % g.lpv, vslot = threadIdx.x / g.lpv;
  const bool lane_ok = vslot < g.vs;
  const int C = g.C;
  for (int i = threadIdx.x; i < K * K * K * C; i += 256) lds[i] = w[i];
  __syncthreads();
  float s1[VEC], s2[VEC], bv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; bv[i] = bias ? bias[cv * VEC + i] : 0.f; }
  const int GX = (g.W + XB - 1) / XB;
  const long groups = (long)g.D * g.H * GX;
  const T* xn = x + (long)n * g.D * g.H * g.W * C + cv * VEC;
  T* yn = y + (long)n * g.D * g.H * g.W * C + cv * VEC;
  const float* wl = lds + cv * VEC;
  for (int it = 0; it < g.iters; ++it) {
    const long G = ((long)slot * g.iters + it) * g.vs + vslot;
    if (!lane_ok || G >= groups) continue;
    const int gx = (int)(G % GX);
    const long t = G / GX;
    const int oy = (int)(t % g.H), oz = (int)(t / g.H);
    const int x0 = gx * XB;
    int xi[NI];
    bool xv[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) { const int ix = x0 - P + k; xv[k] = ix >= 0 && ix < g.W; xi[k] = min(max(ix, 0), g.W - 1); }
    float acc[XB][VEC];
#pragma unroll
    for (int v = 0; v < XB; ++v)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[v][i] = bv[i];
    for (int kz = 0; kz < K; ++kz) {
      const int iz = oz - P + kz;
      const bool zok = iz >= 0 && iz < g.D;
      const int zc = min(max(iz, 0), g.D - 1);
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy - P + ky;
        const bool zy = zok && iy >= 0 && iy < g.H;
        const T* row = xn + ((long)zc * g.H + min(max(iy, 0), g.H - 1)) * g.W * C;
        float in[NI][VEC];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
          VecIO<T, VEC>::load(row + (long)xi[k] * C, in[k]);
          const bool ok = zy & xv[k];                    // zero padding: the VALUE is zeroed (the tap is shared by XB outputs)
#pragma unroll
          for (int i = 0; i < VEC; ++i) in[k][i] = ok ? in[k][i] : 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          float wv[VEC];
          VecIO<float, VEC>::load(wl + ((kz * K + ky) * K + kx) * C, wv);
#pragma unroll
          for (int v = 0; v < XB; ++v)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[v][i] = fmaf(in[v + kx][i], wv[i], acc[v][i]);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < XB; ++v) {
      if (x0 + v >= g.W) continue;
      VecIO<T, VEC>::store(yn + (((long)oz * g.H + oy) * g.W + x0 + v) * C, acc[v]);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float r = to_f32<T>(from_f32<T>(acc[v][i]));
        s1[i] += r;
        s2[i] = fmaf(r, r, s2[i]);
      }
    }
  }
  __syncthreads();
  if (stats) block_stats_reduce<VEC>(s1, s2, cv, vslot, lane_ok, g, stats, n, slot, lds);
}

// ---------------------------------------------------------------------------------------------
// Depthwise transposed conv, K = 3, stride 2, pad 1, "cell" form.  The padded output grid (2D,2H,2W) splits into D*H*W
// cells of 2x2x2 positions p = 2m + {0,1}; with o = p - 1 the even positions read inputs (m-1: tap 2, m: tap 0) and the
// odd ones input m (tap 1), so a lane loads the 2x2x2 inputs m-1..m once (16 bytes each) and produces all 8 outputs:
// no parity divergence, 8 loads per 8 stores, the 27 x C taps staged once per workgroup in LDS (the lanes of a
// channel chunk read the same address: broadcast).  m-1 = -1 only feeds p = 0 positions, which are the zero faces.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
dwconvT3d_k3_cell_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w,
                         const float* __restrict__ bias, float* __restrict__ stats, DwGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 27*C taps, later the statistics scratch
  const int n = blockIdx.y, slot = blockIdx.x;
  const int cv = threadIdx.x % g.lpv, vslot = threadIdx.x / g.lpv;
  const bool lane_ok = vslot < g.vs;
  const int C = g.C;
  for (int i = threadIdx.x; i < 27 * C; i += 256) lds[i] = w[i];
  __syncthreads();
  float s1[VEC], s2[VEC], bv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; bv[i] = bias ? bias[cv * VEC + i] : 0.f; }
  const long vcells = (long)g.D * g.H * g.W;
  const T* xn = x + (long)n * vcells * C + cv * VEC;
  T* yn = y + (long)n * g.Do * g.Ho * g.Wo * C + cv * VEC;
  const float* wl = lds + cv * VEC;
  for (int it = 0; it < g.iters; ++it) {
    const long c = ((long)slot * g.iters + it) * g.vs + vslot;
    if (!lane_ok || c >= vcells) continue;
    const int mx = (int)(c % g.W);
    const long t = c / g.W;
    const int my = (int)(t % g.H), mz = (int)(t / g.H);
    float xin[2][2][2][VEC];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int iz = max(mz - 1 + a, 0), iy = max(my - 1 + b, 0), ix = max(mx - 1 + d, 0);
          VecIO<T, VEC>::load(xn + (((long)iz * g.H + iy) * g.W + ix) * C, xin[a][b][d]);
        }
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int Pz = 2 * mz + pz, Py = 2 * my + py, Px = 2 * mx + px;
          float acc[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = bv[i];
          // per axis: even position -> (input 0, tap 2), (input 1, tap 0); odd position -> (input 1, tap 1)
#pragma unroll
          for (int a = pz; a < 2; ++a)
#pragma unroll
            for (int b = py; b < 2; ++b)
#pragma unroll
              for (int d = px; d < 2; ++d) {
                const int kz = pz ? 1 : (a ? 0 : 2), ky = py ? 1 : (b ? 0 : 2), kx = px ? 1 : (d ? 0 : 2);
                float wv[VEC];
                VecIO<float, VEC>::load(wl + ((kz * 3 + ky) * 3 + kx) * C, wv);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = fmaf(xin[a][b][d][i], wv[i], acc[i]);
              }
          const bool face = (Pz == 0) | (Py == 0) | (Px == 0);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = face ? 0.f : acc[i];
          // y == nullptr: statistics only (the fused up-block mixer recomputes t in its prologue, pw_mlp_up_kernels.hip)
          if (y) VecIO<T, VEC>::store(yn + (((long)Pz * g.Ho + Py) * g.Wo + Px) * C, acc);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float r = to_f32<T>(from_f32<T>(acc[i]));     // faces contribute exact zeros
            s1[i] += r;
            s2[i] = fmaf(r, r, s2[i]);
          }
        }
  }
  __syncthreads();                                               // the taps are dead: the scratch may reuse their space
  if (stats) block_stats_reduce<VEC>(s1, s2, cv, vslot, lane_ok, g, stats, n, slot, lds);
}

// ---------------------------------------------------------------------------------------------
// Fast path: K = 3, stride 1 -- "z-march".  One workgroup owns an 8 x 8 (y,x) footprint of a 32-channel
// group and marches along z.  Per step ONE haloed input plane (10 x 10 x 32 ch) is staged in LDS as fp32
// (converted once, double buffered; the global loads of the next 2-3 planes are in flight in registers, issued as
// inline asm and awaited with a COUNTED s_waitcnt so that younger planes survive the wait); each thread owns
// (4 positions, 2 channels), reads the 27 x 32 taps from LDS (broadcast) and carries the z extent of the stencil in
// three rolling fp32 accumulators (outputs z-1, z, z+1), so
//   * every LDS data value (one volatile ds_read_b64 per tap and position) feeds 3 packed FMAs,
//   * the tap loop is hand scheduled: reads one tap group ahead, 12 asm v_pk_fma_f32 round-robin over the 12
//     independent accumulators -- 120 VGPRs, 4 workgroups per CU,
//   * HBM sees x once (+ the y/x halo) and y once.
// Statistics (sum / sum of squares of the stored values) leave as one partial per workgroup.
constexpr int TILE_Y = 8, TILE_X = 8;
constexpr int MARCH_CG = 32;

// struct DwMarch: dwconv_march.h (shared with dwconv_mfma_kernels.hip)

// RES: y = conv(x) + res (res laid out like y).  The training backward uses it for the data gradient of a residual block,
// dx = conv_reversed(dt) + dy: the separate read-modify-write pass over dx (add_inplace: 3 tensor passes, 1.8 ms of a
// 36 ms MedNeXt-S step) becomes one extra read inside the kernel that already writes dx.  The residual values of output
// plane p travel in registers (one channel pair per position: 4-byte loads, 64-B segments), requested at step p -- BEFORE
// that step's input-plane request -- and consumed by the stores of step p+1; the counted wait below includes them.
// H16 (bf16 storage only): the nine in-plane taps of a z step are accumulated in PACKED f16 (v_pk_fma_f16: two channels per
// lane-instruction at full rate, where v_pk_fma_f32 takes two passes) and each 9-tap partial sum is then added to its fp32
// accumulator.  The kernel is bound by VALU issue -- 27 FMAs x 360 M outputs per level-0 launch = 250 us at the fp32 FMA peak,
// 340 us as scheduled -- so this removes ~40 % of its issue cycles; the input plane and the taps live in LDS as f16 (half the
// bytes per read).  Error budget: the bf16 inputs are exact in f16 (8 -> 11 mantissa bits), a 9-term f16 partial sum carries
// ~sqrt(9) * 2^-12 = 7e-4 relative error, below the 2^-9 rounding of the bf16 result; the z direction stays in fp32.
// TX / NT: x extent of the footprint and threads per workgroup.  8 / 256 is the original shape; 16 / 512 (two 8 x 8 sub-tiles staged
// as ONE 10 x 18 haloed plane, same work and registers per thread) shares the halo columns between the sub-tiles inside the
// workgroup instead of hoping for an L2 hit: x-halo traffic 10/8 -> 18/16, and a row segment of 18 voxels touches 10 cache
// lines for 8 useful ones where two 10-voxel segments touch 12.
template <typename T, int VEC, int PF, bool ASYNC, int WPS = 2, bool RES = false, bool H16 = false, int TX = 8, int NT = 256>
__global__ void __launch_bounds__(NT, WPS)
dwconv3d_k3_march_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w,
                         const float* __restrict__ bias, float* __restrict__ stats, DwMarch g,
                         const T* __restrict__ res = nullptr) {
  // VEC channels per lane (4: ds_read_b128, 108 weight registers; 2: ds_read_b64, 54 weight registers ->
  // more resident workgroups).  PF = planes of global loads kept in flight (register staged).
  constexpr int TILE_X = TX;
  constexpr int CG = MARCH_CG, LPV = CG / VEC, PPP = NT / LPV, PASSES = (TILE_Y * TILE_X) / PPP;
  constexpr int EY = TILE_Y + 2, EX = TILE_X + 2;
  constexpr int EPC = 16 / (int)sizeof(T);          // elements per 16-byte chunk
  constexpr int CH16 = CG / EPC;                    // chunks per voxel
  constexpr int NCHUNK = EY * EX * CH16;            // chunks per plane
  constexpr int CPT = (NCHUNK + NT - 1) / NT;       // chunks per thread
  typedef float fvec_t __attribute__((ext_vector_type(VEC)));
  static_assert(!H16 || (sizeof(T) == 2 && VEC == 2), "the packed-f16 tap loop is for bf16 storage, channel pairs");
  typedef typename std::conditional<H16, _Float16, float>::type lds_t;
  __shared__ __attribute__((aligned(16))) lds_t plane[2][EY * EX * CG];
  __shared__ float red[NT / 64][2][CG];
  // the 27 x CG taps live in LDS, not in 54 registers per lane (lanes of one channel pair read the same address: broadcast)
  __shared__ __attribute__((aligned(16))) lds_t wlds[27 * CG];

  const int tid = threadIdx.x;
  // 1-D grid, XCD-aware: logical index = ((n * CGs + cg) * slots + slot); x-/y-neighbouring footprints
  // (which share halo columns) are consecutive logical indices -> same XCD, same L2
  int b = g.swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  // channel groups innermost: the workgroups that split a voxel's channels (64 bytes each of its 128-byte lines) are dispatched back
  // to back on one XCD, so the second one finds the lines in L2 (slot-major order fetched every line of a C >= 64 level twice:
  // 56^3 x 64 at 2.4 TB/s against 3.4 at C = 32, profiles/r04_dwconv_mfma.txt)
  const int ncg = g.C / MARCH_CG;
  const int cg = g.cg_inner ? b % ncg : (b / g.slots) % ncg;
  const int slot_id = g.cg_inner ? (b / ncg) % g.slots : b % g.slots;
  const int n = b / (ncg * g.slots);
  b = slot_id;
  const int fx = b % g.tx; b /= g.tx;
  const int fy = b % g.ty;
  const int zchunk = b / g.ty;
  const int y0 = fy * TILE_Y, x0 = fx * TILE_X;
  const int zs = zchunk * g.zc;
  const int ze = min(zs + g.zc, g.D);               // outputs [zs, ze)
  const int C = g.C;
  const long plane_elems = (long)g.H * g.W * C;
  const T* xn = x + (long)n * g.D * plane_elems + cg * CG;
  T* yn = y + (long)n * g.D * plane_elems + cg * CG;
  const T* rn = RES ? res + (long)n * g.D * plane_elems + cg * CG : nullptr;

  // ---- staging descriptors (constant along z)
  int goff[CPT], loff[CPT];
  bool cok[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + NT * i;
    const int vox = c / CH16, part = c % CH16;
    const int yy = vox / EX, xx = vox % EX;
    const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
    cok[i] = (c < NCHUNK) && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
    // every lane always loads (clamped address, zero-filled at commit): no branches around the loads, and with ASYNC
    // each wave issues exactly CPT loads per plane, which makes the counted s_waitcnt below exact
    const int gyc = min(max(gy, 0), g.H - 1), gxc = min(max(gx, 0), g.W - 1);
    goff[i] = (gyc * g.W + gxc) * C + part * EPC;
    loff[i] = (c < NCHUNK) ? vox * CG + part * EPC : -1;
  }
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t stg0[CPT], stg1[CPT], stg2[PF == 3 ? CPT : 1];
  // ASYNC: the plane loads are inline asm, invisible to the compiler's wait-count pass (which, with stores pending on the
  // same counter, would drain everything with vmcnt(0) at the first use).  Loads return in order among loads, so waiting
  // until at most NEWER = CPT*(planes issued later) operations are outstanding guarantees this plane has landed
  // whatever the stores do -- the younger planes stay in flight across the barrier.
  auto issue = [&](int gz, u32x4_t (&stg)[CPT]) {
    const int zc = min(max(gz, 0), g.D - 1);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const T* ptr = xn + (long)zc * plane_elems + goff[i];
      if constexpr (ASYNC) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(stg[i]) : "v"(ptr) : "memory");
      else stg[i] = *reinterpret_cast<const u32x4_t*>(ptr);
    }
  };
  // with_res: this step also issued its PASSES residual loads (before its plane request), and the previous step's residual
  // values are about to be used: everything younger than THOSE may stay in flight -- the previous step's plane request, this
  // step's residual loads, this step's plane request (PF = 3; with PF = 2 the awaited plane is itself that younger request)
  auto landed = [&](u32x4_t (&stg)[CPT], bool younger_in_flight, bool with_res) {
    if constexpr (ASYNC) {
      static_assert(CPT == 2, "the wait below names two staging registers");
      // operand-free waits (a "+v" operand here made hipcc copy the still-in-flight registers BEFORE the wait), then one
      // anchor that orders every later use of the staging registers after them
      if (!younger_in_flight) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
      else if (RES && with_res) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CPT * (PF - 1) + PASSES) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CPT * (PF - 1)) : "memory");
      asm volatile("" : "+v"(stg[0]), "+v"(stg[1]) : : "memory");
    }
  };
  auto commit = [&](int slot, u32x4_t (&stg)[CPT], int gz) {
    const bool zok = gz >= 0 && gz < g.D;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      if (loff[i] < 0) continue;
      float v[EPC];
      VecIO<T, EPC>::load(reinterpret_cast<const T*>(&stg[i]), v);
      if (!(zok && cok[i])) {
#pragma unroll
        for (int q = 0; q < EPC; ++q) v[q] = 0.f;
      }
      lds_t* dst = &plane[slot][loff[i]];
      if constexpr (H16) {
        // f16 image: clamp into the f16 range first (a bf16 activation beyond it would become inf and poison the sums --
        // v_cvt_pkrtz_f16_f32 does NOT saturate on gfx950, measured in round 4), as ONE v_med3_f32 per element: fminf(fmaxf())
        // compiled to v_max + v_med3 (32 of the step's ~226 VALU instructions for 16 elements).  Every bf16 value inside the normal
        // f16 range converts exactly whatever the rounding mode, so the packed round-toward-zero conversion is as good as RNE here.
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4 packed;
#pragma unroll
        for (int q = 0; q < EPC; q += 2)
          packed[q / 2] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_fmed3f(v[q], -60000.f, 60000.f),
                                                                                      __builtin_amdgcn_fmed3f(v[q + 1], -60000.f, 60000.f)));
        *reinterpret_cast<u32x4*>(dst) = packed;
      } else {
#pragma unroll
        for (int q = 0; q < EPC; q += 4)
          *reinterpret_cast<f32x4_t*>(dst + q) = f32x4_t{v[q], v[q + 1], v[q + 2], v[q + 3]};
      }
    }
  };

  // ---- per-thread weights (27 taps x VEC channels), bias, positions
  const int cv = tid % LPV, pslot = tid / LPV;
  const int c0 = cg * CG + cv * VEC;
  for (int i = tid; i < 27 * CG; i += NT) wlds[i] = (lds_t)w[(long)(i / CG) * C + cg * CG + (i % CG)];
  // (visible after the __syncthreads() that follows the prologue's first commit)
  float bv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) bv[i] = bias ? bias[c0 + i] : 0.f;
  int lbase[PASSES];
  long obase[PASSES];
  bool pok[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int pos = ps * PPP + pslot;
    const int py = pos / TILE_X, px = pos % TILE_X;
    lbase[ps] = (py * EX + px) * CG + cv * VEC;
    pok[ps] = (y0 + py) < g.H && (x0 + px) < g.W;
    obase[ps] = ((long)(y0 + py) * g.W + (x0 + px)) * C + cv * VEC;
  }
  // residual staging: one 4-byte (bf16 pair) / 8-byte (fp32 pair) value per position; three rotating sets, two live
  typedef typename std::conditional<sizeof(T) == 2, unsigned int, unsigned long long>::type rq_t;
  rq_t rq0[RES ? PASSES : 1], rq1[RES ? PASSES : 1], rq2[RES ? PASSES : 1];
  auto issue_res = [&](int gz, rq_t (&rq)[RES ? PASSES : 1]) {
    if constexpr (RES) {
      static_assert(VEC == 2, "residual staging carries channel pairs");
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        // every lane always loads (clamped position): exactly PASSES loads per wave and step keep the counted wait exact
        const int pos = ps * PPP + pslot;
        const int yy = min(y0 + pos / TILE_X, g.H - 1), xx = min(x0 + pos % TILE_X, g.W - 1);
        const T* ptr = rn + (long)gz * plane_elems + ((long)yy * g.W + xx) * C + cv * VEC;
        if constexpr (ASYNC) {
          if constexpr (sizeof(T) == 2) asm volatile("global_load_dword %0, %1, off" : "=&v"(rq[ps]) : "v"(ptr) : "memory");
          else asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(rq[ps]) : "v"(ptr) : "memory");
        } else {
          rq[ps] = *reinterpret_cast<const rq_t*>(ptr);
        }
      }
    }
  };
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }

  fvec_t accA[PASSES], accB[PASSES], accC[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
    for (int i = 0; i < VEC; ++i) { accA[ps][i] = bv[i]; accB[ps][i] = bv[i]; accC[ps][i] = bv[i]; }

  // one z step: input plane gz lives in plane[slot]; prev/cur/next = outputs gz-1 / gz / gz+1.
  // `ld` receives the global loads issued this step (plane gz+PF); `cm` holds plane gz+1 (issued PF-1
  // steps ago) and is committed to the other LDS slot after the compute.
  auto step = [&](int gz, int slot, fvec_t (&prev)[PASSES], fvec_t (&cur)[PASSES],
                  fvec_t (&next)[PASSES], u32x4_t (&ld)[CPT], u32x4_t (&cm)[CPT],
                  rq_t (&rnew)[RES ? PASSES : 1], rq_t (&rold)[RES ? PASSES : 1]) {
    const bool res_now = RES && gz >= zs && gz < ze;      // residual of output plane gz (stored by the NEXT step)
    if (res_now) issue_res(gz, rnew);
    if (gz + PF <= ze) issue(gz + PF, ld);
    // Unconditional accumulation: planes outside the volume were staged as zeros, and accumulators that
    // belong to outputs outside [zs, ze) are simply never stored (2 wasted planes per z-chunk), which keeps
    // the inner loop free of per-FMA selects.
    if constexpr (H16) {
      // packed-f16 form of the hand-scheduled loop below: tap group g+1 (3 weight + 4 data reads, 4 bytes each) is requested
      // before the 12 v_pk_fma_f16 of group g; the three 9-tap partial sums per position then join the fp32 accumulators
      typedef const volatile __attribute__((address_space(3))) h2_t* lds_vol_h2;
      h2_t wq[2][3], vq[2][PASSES];
      h2_t pn[PASSES], pc[PASSES], pp[PASSES];
      auto fetch = [&](int g, int buf) {
        const int dy = g / 3, dx = g % 3;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) wq[buf][kz] = *(lds_vol_h2)(&wlds[((kz * 3 + dy) * 3 + dx) * CG + cv * VEC]);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) vq[buf][ps] = *(lds_vol_h2)(&plane[slot][lbase[ps] + (dy * EX + dx) * CG]);
      };
      fetch(0, 0);
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        if (g + 1 < 9) fetch(g + 1, (g + 1) & 1);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          if (g == 0) {
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(pn[ps]) : "v"(vq[0][ps]), "v"(wq[0][0]));
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(pc[ps]) : "v"(vq[0][ps]), "v"(wq[0][1]));
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(pp[ps]) : "v"(vq[0][ps]), "v"(wq[0][2]));
          } else {
            asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(pn[ps]) : "v"(vq[g & 1][ps]), "v"(wq[g & 1][0]));
            asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(pc[ps]) : "v"(vq[g & 1][ps]), "v"(wq[g & 1][1]));
            asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(pp[ps]) : "v"(vq[g & 1][ps]), "v"(wq[g & 1][2]));
          }
        }
      }
      // acc (fp32) += partial (f16 half): one v_fma_mix_f32 each (f16 source selected by op_sel, times 1.0, plus the fp32 acc)
#define PYTC_MIX_ADD(ACC, P)                                                                                                  \
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(ACC[0]) : "v"(P));                     \
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ACC[1]) : "v"(P));
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        float an[2] = {next[ps][0], next[ps][1]}, ac[2] = {cur[ps][0], cur[ps][1]}, ap[2] = {prev[ps][0], prev[ps][1]};
        PYTC_MIX_ADD(an, pn[ps]) PYTC_MIX_ADD(ac, pc[ps]) PYTC_MIX_ADD(ap, pp[ps])
        next[ps][0] = an[0]; next[ps][1] = an[1]; cur[ps][0] = ac[0]; cur[ps][1] = ac[1]; prev[ps][0] = ap[0]; prev[ps][1] = ap[1];
      }
#undef PYTC_MIX_ADD
    } else {
      // Hand-scheduled tap loop (VEC == 2).  Left to itself hipcc issues all 63 LDS reads of a step first (126 live
      // registers) and then runs each accumulator's nine FMAs back to back, with an s_nop between dependent
      // v_pk_fma_f32.  Here the order is fixed by volatile reads and asm-volatile FMAs: tap group g+1 (3 weight + 4 data
      // reads) is requested before the 12 FMAs of group g, which go round-robin over the 12 independent accumulators
      // (a dependent FMA is 12 issue slots away).
      static_assert(VEC == 2, "hand-scheduled path is written for channel pairs");
      typedef const volatile __attribute__((address_space(3))) fvec_t* lds_vol_ptr;
      fvec_t wq[2][3], vq[2][PASSES];
      auto fetch = [&](int g, int buf) {
        const int dy = g / 3, dx = g % 3;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) wq[buf][kz] = *(lds_vol_ptr)(&wlds[((kz * 3 + dy) * 3 + dx) * CG + cv * VEC]);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) vq[buf][ps] = *(lds_vol_ptr)(&plane[slot][lbase[ps] + (dy * EX + dx) * CG]);
      };
      fetch(0, 0);
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        if (g + 1 < 9) fetch(g + 1, (g + 1) & 1);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(next[ps]) : "v"(vq[g & 1][ps]), "v"(wq[g & 1][0]));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(cur[ps]) : "v"(vq[g & 1][ps]), "v"(wq[g & 1][1]));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(prev[ps]) : "v"(vq[g & 1][ps]), "v"(wq[g & 1][2]));
        }
      }
    }
    if (gz + 1 <= ze) {
      landed(cm, gz + PF <= ze, res_now);
      commit(slot ^ 1, cm, gz + 1);
    } else if constexpr (ASYNC) {
      // last step: the residual of plane ze-1 has to be there (RES).  Without a residual nothing is in flight here; the statement stays
      // so that EVERY path through a step executes a hand-written wait after the step's requests (csrc/asm_check.py, build-time)
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    }
    if constexpr (RES && ASYNC) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) asm volatile("" : "+v"(rold[ps]) : : "memory");    // uses stay after the wait
    }
    // stores after the wait: the only operations younger than the awaited plane are then the newest plane's loads
    if (gz - 1 >= zs) {   // output plane gz-1 is complete
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        if (pok[ps]) {
          float pv[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) pv[i] = prev[ps][i];
          if constexpr (RES) {
            if constexpr (sizeof(T) == 2) {
              pv[0] += __uint_as_float((unsigned int)rold[ps] << 16);
              pv[1] += __uint_as_float((unsigned int)rold[ps] & 0xffff0000u);
            } else {
              pv[0] += __uint_as_float((unsigned int)rold[ps]);
              pv[1] += __uint_as_float((unsigned int)((unsigned long long)rold[ps] >> 32));
            }
          }
          VecIO<T, VEC>::store(yn + (long)(gz - 1) * plane_elems + obase[ps], pv);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float r = to_f32<T>(from_f32<T>(prev[ps][i]));
            s1[i] += r;
            s2[i] = fmaf(r, r, s2[i]);
          }
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
      for (int i = 0; i < VEC; ++i) prev[ps][i] = bv[i];
    __syncthreads();
  };

  // prologue: plane zs-1 -> LDS slot 0; with PF >= 2 planes zs (.. zs+1) are already in flight in stg1 (stg2)
  issue(zs - 1, stg0);
  if (PF >= 2) issue(zs, stg1);
  if constexpr (PF == 3) issue(zs + 1, stg2);
  landed(stg0, true, false);
  commit(0, stg0, zs - 1);
  __syncthreads();
  int slot = 0;
  // residual sets rotate with period 3 like the accumulators: step k requests into set k % 3 and consumes set (k - 1) % 3
  if constexpr (PF == 3) {
    // plane p travels in set (p - (zs-1)) % 3: step k loads plane gz+3 into set k%3, commits plane gz+1 from set (k+1)%3
    for (int gz = zs - 1; gz <= ze; gz += 3) {
      step(gz, slot, accA, accB, accC, stg0, stg1, rq0, rq2); slot ^= 1;
      if (gz + 1 <= ze) { step(gz + 1, slot, accB, accC, accA, stg1, stg2, rq1, rq0); slot ^= 1; }
      if (gz + 2 <= ze) { step(gz + 2, slot, accC, accA, accB, stg2, stg0, rq2, rq1); slot ^= 1; }
    }
  } else if (PF == 1) {
    for (int gz = zs - 1; gz <= ze; gz += 3) {
      step(gz, slot, accA, accB, accC, stg0, stg0, rq0, rq2); slot ^= 1;
      if (gz + 1 <= ze) { step(gz + 1, slot, accB, accC, accA, stg0, stg0, rq1, rq0); slot ^= 1; }
      if (gz + 2 <= ze) { step(gz + 2, slot, accC, accA, accB, stg0, stg0, rq2, rq1); slot ^= 1; }
    }
  } else {
    // plane p travels in stg[(p - (zs-1)) & 1]: step k (gz = zs-1+k) loads plane gz+2 into set k&1 and
    // commits plane gz+1 from set (k+1)&1
    for (int gz = zs - 1; gz <= ze; gz += 6) {
      step(gz, slot, accA, accB, accC, stg0, stg1, rq0, rq2); slot ^= 1;
      if (gz + 1 <= ze) { step(gz + 1, slot, accB, accC, accA, stg1, stg0, rq1, rq0); slot ^= 1; }
      if (gz + 2 <= ze) { step(gz + 2, slot, accC, accA, accB, stg0, stg1, rq2, rq1); slot ^= 1; }
      if (gz + 3 <= ze) { step(gz + 3, slot, accA, accB, accC, stg1, stg0, rq0, rq2); slot ^= 1; }
      if (gz + 4 <= ze) { step(gz + 4, slot, accB, accC, accA, stg0, stg1, rq1, rq0); slot ^= 1; }
      if (gz + 5 <= ze) { step(gz + 5, slot, accC, accA, accB, stg1, stg0, rq2, rq1); slot ^= 1; }
    }
  }
  if constexpr (ASYNC) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // (asm_check.py: the epilogue is reached through a wait)

  if (stats) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
#pragma unroll
      for (int off = LPV; off < 64; off <<= 1) {
        s1[i] += __shfl_xor(s1[i], off, 64);
        s2[i] += __shfl_xor(s2[i], off, 64);
      }
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < LPV) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        red[wave][0][lane * VEC + i] = s1[i];
        red[wave][1][lane * VEC + i] = s2[i];
      }
    }
    __syncthreads();
    if (tid < 2 * CG) {
      const int which = tid / CG, ch = tid % CG;
      float a = 0.f;
#pragma unroll
      for (int wv = 0; wv < NT / 64; ++wv) a += red[wv][which][ch];
      stats[(((long)n * g.slots + slot_id) * 2 + which) * C + cg * CG + ch] = a;
    }
  }
}

static bool march_ok(int D, int H, int W, int C, int K, int stride, int dtype, int transposed) {
  if (transposed || K != 3 || stride != 1) return false;
  if (C % MARCH_CG) return false;
  return D >= 8 && H >= 16 && W >= 16 && (long)H * W * C < (1L << 30);
}

// planes too small for the VALU z-march (8 <= H, W < 16; 14^3 at level 3 of a 112^3 window) still fit the matrix-core form, whose
// footprints handle ragged tiles the same way: bf16 forward launches only (the gather / x-block kernels keep everything else)
static bool mfma_small_ok(int D, int H, int W, int C, int K, int stride, int dtype, int transposed) {
  if (transposed || K != 3 || stride != 1 || dtype != PYTC_BF16 || (C % MARCH_CG) != 0) return false;
  if (tuning_get("dwconv_mfma", 1) == 0 || tuning_get("dwconv_mfma_small", 1) == 0) return false;
  return D >= 8 && H >= 8 && W >= 8 && (H < 16 || W < 16);
}

// the forward kernel's footprint: 8 x 16 where the rows divide into whole 16-voxel tiles (level 0: W = 112), 8 x 8 otherwise
static int march_tile_x(int W, int dtype) {
  return (dtype == PYTC_BF16 && W % 16 == 0 && tuning_get("dwconv_march_tx16", 0) != 0 && tuning_get("dwconv_mfma", 1) == 0) ? 16 : TILE_X;
}

static void make_march(DwMarch& t, int N, int D, int H, int W, int C, int tilex = TILE_X) {
  t.N = N; t.D = D; t.H = H; t.W = W; t.C = C;
  t.tilex = tilex;
  t.ty = (H + TILE_Y - 1) / TILE_Y; t.tx = (W + tilex - 1) / tilex;
  // z-chunks: enough workgroups PER SAMPLE to fill the chip (>= 1024: 4 per CU at batch 1), chunks >= 14 planes (halo <= 14 %).
  // The split must not depend on N: the statistics partials (one per workgroup) are summed in slot order, so a sample's
  // mean / rstd -- and with them its bf16 prediction -- would otherwise change with the batch it happens to travel in
  // (chunked == whole-volume and rank-sharded == single-process exactness rely on batch-invariant windows).
  const long fp = (long)t.ty * t.tx * (C / MARCH_CG);
  // Measured and removed (round 4): choosing the split that minimises ceil(workgroups / 1024 resident) x (zc + 2) for an 8-window batch
  // (56^3 x 64: 5 chunks instead of 4; 112^3 x 32: 7 instead of 6) changed no launch time (396 / 116 / 41 us either way): the
  // workgroups of these kernels do not advance in rounds, the launch is throughput bound.
  int nzc = (int)(((long)tuning_get("dwconv_march_wgs", 1024) + fp - 1) / fp);
  if (nzc < 1) nzc = 1;
  // ... unless 14-plane chunks leave a sample with fewer than `dwconv_march_small_wgs` workgroups (the 20^3 x 256 level of MedNeXt-L: 72): such a
  // launch is a chain of D + 2 dependent plane steps on a fraction of the chip, its bytes are irrelevant -- chunks of >= 5 planes then
  // (24.5 -> 16.5 us there; applied to the larger levels the extra halo costs more than it saves: profiles/r05_short_z_chunks.txt).
  // A per-sample rule: the split still does not depend on N.
  const int min_planes = fp * (D / 14 > 0 ? D / 14 : 1) < (long)tuning_get("dwconv_march_small_wgs", 128) ? 5 : 14;
  int maxc = D / tuning_get("dwconv_march_min_planes", min_planes);
  if (maxc < 1) maxc = 1;
  if (nzc > maxc) nzc = maxc;
  t.zc = (D + nzc - 1) / nzc;
  t.nzc = (D + t.zc - 1) / t.zc;
  t.slots = t.ty * t.tx * t.nzc;
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3x3 stride-1 weight gradient, z-march form:  dW[kz][ky][kx][c] = sum_o G[o][c] * X[o + k - 1][c].
// Same footprint / staging as the forward march (X plane with halo -> fp32 LDS image, one plane per step), but the
// roles flip: the 27 x 2 ACCUMULATORS are the taps (per lane: one channel pair), and the three rolling register sets
// hold G at planes gz-1 / gz / gz+1 for the lane's four positions: X plane gz meets G[gz+1] in the kz=0 taps, G[gz] in
// kz=1, G[gz-1] in kz=2, so each LDS read feeds three packed FMAs.  G comes straight from HBM one plane ahead (4-byte
// channel-pair loads, 64-B segments per position).  Partials: lanes of equal channel pair are summed by two xor
// shuffles, the four waves in LDS in wave order -> dWp[slot][27][C], dbp[slot][C]; reduce_slots finishes.
template <typename T>
__global__ void __launch_bounds__(256, 2)
dw_wgrad_march_kernel(const T* __restrict__ gr, const T* __restrict__ x, float* __restrict__ dWp,
                      float* __restrict__ dbp, DwMarch g) {
  constexpr int VEC = 2, CG = MARCH_CG, LPV = CG / VEC, PPP = 256 / LPV, PASSES = (TILE_Y * TILE_X) / PPP;
  constexpr int EY = TILE_Y + 2, EX = TILE_X + 2;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int CH16 = CG / EPC;
  constexpr int NCHUNK = EY * EX * CH16;
  constexpr int CPT = (NCHUNK + 255) / 256;
  typedef float fvec_t __attribute__((ext_vector_type(VEC)));
  __shared__ __attribute__((aligned(16))) float plane[2][EY * EX * CG];
  static_assert(4 * 28 * CG <= 2 * EY * EX * CG, "cross-wave scratch must fit in the plane buffers");

  const int tid = threadIdx.x;
  int b = g.swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  // channel groups innermost: the workgroups that split a voxel's channels (64 bytes each of its 128-byte lines) are dispatched back
  // to back on one XCD, so the second one finds the lines in L2 (slot-major order fetched every line of a C >= 64 level twice:
  // 56^3 x 64 at 2.4 TB/s against 3.4 at C = 32, profiles/r04_dwconv_mfma.txt)
  const int ncg = g.C / MARCH_CG;
  const int cg = g.cg_inner ? b % ncg : (b / g.slots) % ncg;
  const int slot_id = g.cg_inner ? (b / ncg) % g.slots : b % g.slots;
  const int n = b / (ncg * g.slots);
  b = slot_id;
  const int fx = b % g.tx; b /= g.tx;
  const int fy = b % g.ty;
  const int zchunk = b / g.ty;
  const int y0 = fy * TILE_Y, x0 = fx * TILE_X;
  const int zs = zchunk * g.zc;
  const int ze = min(zs + g.zc, g.D);               // G planes [zs, ze) belong to this workgroup
  const int C = g.C;
  const long plane_elems = (long)g.H * g.W * C;
  const T* xn = x + (long)n * g.D * plane_elems + cg * CG;
  const T* gn = gr + (long)n * g.D * plane_elems + cg * CG;

  int goff[CPT], loff[CPT];
  bool cok[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + 256 * i;
    const int vox = c / CH16, part = c % CH16;
    const int yy = vox / EX, xx = vox % EX;
    const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
    cok[i] = (c < NCHUNK) && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
    // every lane always loads from a clamped address (zero-filled at commit): no branch around a load, so hipcc can
    // count its waits and the loads of two steps stay in flight
    goff[i] = (min(max(gy, 0), g.H - 1) * g.W + min(max(gx, 0), g.W - 1)) * C + part * EPC;
    loff[i] = (c < NCHUNK) ? vox * CG + part * EPC : -1;
  }
  uint4 stgA[CPT], stgB[CPT];
  auto issue = [&](int gz, uint4 (&stg)[CPT]) {
    const int zc = min(max(gz, 0), g.D - 1);
#pragma unroll
    for (int i = 0; i < CPT; ++i) stg[i] = *reinterpret_cast<const uint4*>(xn + (long)zc * plane_elems + goff[i]);
  };
  auto commit = [&](int slot, uint4 (&stg)[CPT], int gz) {
    const bool zok = gz >= 0 && gz < g.D;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      if (loff[i] < 0) continue;
      float v[EPC];
      VecIO<T, EPC>::load(reinterpret_cast<const T*>(&stg[i]), v);
      if (!(zok && cok[i])) {
#pragma unroll
        for (int q = 0; q < EPC; ++q) v[q] = 0.f;
      }
      float* dst = &plane[slot][loff[i]];
#pragma unroll
      for (int q = 0; q < EPC; q += 4)
        *reinterpret_cast<f32x4_t*>(dst + q) = f32x4_t{v[q], v[q + 1], v[q + 2], v[q + 3]};
    }
  };

  const int cv = tid % LPV, pslot = tid / LPV;
  int lbase[PASSES];
  long obase[PASSES];
  bool pok[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int pos = ps * PPP + pslot;
    const int py = pos / TILE_X, px = pos % TILE_X;
    lbase[ps] = (py * EX + px) * CG + cv * VEC;
    pok[ps] = (y0 + py) < g.H && (x0 + px) < g.W;
    obase[ps] = ((long)min(y0 + py, g.H - 1) * g.W + min(x0 + px, g.W - 1)) * C + cv * VEC;   // clamped, see pok
  }
  fvec_t acc[27];
  float accb[VEC];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) accb[i] = 0.f;
  // G planes: three converted sets roll through prev / cur / next; two RAW sets hold the loads of G[gz+2] (issued one
  // step ago) and G[gz+3] (issued this step) so that a G load has two steps to land
  typedef typename std::conditional<sizeof(T) == 2, unsigned int, fvec_t>::type graw_t;
  fvec_t gA[PASSES], gB[PASSES], gC[PASSES];
  graw_t grawA[PASSES], grawB[PASSES];
  auto graw_load = [&](int gz, graw_t (&dst)[PASSES]) {
    const int zc = min(max(gz, 0), g.D - 1);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) dst[ps] = *reinterpret_cast<const graw_t*>(gn + (long)zc * plane_elems + obase[ps]);
  };
  auto gconv = [&](int gz, const graw_t (&src)[PASSES], fvec_t (&dst)[PASSES]) {   // 0 outside the chunk / the volume
    const bool zok = gz >= zs && gz < ze;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      fvec_t v;
      if constexpr (sizeof(T) == 2) {
        v[0] = __uint_as_float(src[ps] << 16);
        v[1] = __uint_as_float(src[ps] & 0xffff0000u);
      } else {
        v = src[ps];
      }
      const bool ok = zok && pok[ps];
      dst[ps][0] = ok ? v[0] : 0.f;
      dst[ps][1] = ok ? v[1] : 0.f;
    }
  };
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
    for (int i = 0; i < VEC; ++i) { gA[ps][i] = 0.f; gB[ps][i] = 0.f; }
  graw_load(zs, grawA);
  gconv(zs, grawA, gC);
  graw_load(zs + 1, grawB);

  // one z step: X plane gz is in plane[slot]; prev/cur/next = G[gz-1] / G[gz] / G[gz+1]
  // `ld` / `gl` receive this step's loads (X plane gz+2, G plane gz+3); `cm` / `gc` were loaded one step ago (X plane
  // gz+1, G plane gz+2) and are consumed at the end of the step
  auto step = [&](int gz, int slot, fvec_t (&prev)[PASSES], fvec_t (&cur)[PASSES], fvec_t (&next)[PASSES],
                  uint4 (&ld)[CPT], uint4 (&cm)[CPT], graw_t (&gl)[PASSES], graw_t (&gc)[PASSES]) {
    issue(gz + 2, ld);
    graw_load(gz + 3, gl);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
      for (int i = 0; i < VEC; ++i) accb[i] += next[ps][i];
    // hand-scheduled like the forward march: the four positions' reads of tap (dy,dx)+1 are requested before the 12
    // FMAs of tap (dy,dx); per tap the three kz accumulators take the positions in turn (dependent FMAs 3 slots apart)
    static_assert(VEC == 2, "hand-scheduled path is written for channel pairs");
    {
      typedef const volatile __attribute__((address_space(3))) fvec_t* lds_vol_ptr;
      fvec_t vq[2][PASSES];
      auto fetch = [&](int t, int buf) {
        const int dy = t / 3, dx = t % 3;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) vq[buf][ps] = *(lds_vol_ptr)(&plane[slot][lbase[ps] + (dy * EX + dx) * CG]);
      };
      fetch(0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) fetch(t + 1, (t + 1) & 1);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[0 * 9 + t]) : "v"(vq[t & 1][ps]), "v"(next[ps]));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[1 * 9 + t]) : "v"(vq[t & 1][ps]), "v"(cur[ps]));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2 * 9 + t]) : "v"(vq[t & 1][ps]), "v"(prev[ps]));
        }
      }
    }
    gconv(gz + 2, gc, prev);                                       // the freed set becomes G[gz+2]
    if (gz + 1 <= ze) commit(slot ^ 1, cm, gz + 1);
    __syncthreads();
  };

  issue(zs - 1, stgA);
  issue(zs, stgB);
  commit(0, stgA, zs - 1);
  __syncthreads();
  int slot = 0;
  // accumulator sets rotate with period 3, staging sets with period 2: six steps per trip
  for (int gz = zs - 1; gz <= ze; gz += 6) {
    step(gz, slot, gA, gB, gC, stgA, stgB, grawA, grawB); slot ^= 1;
    if (gz + 1 <= ze) { step(gz + 1, slot, gB, gC, gA, stgB, stgA, grawB, grawA); slot ^= 1; }
    if (gz + 2 <= ze) { step(gz + 2, slot, gC, gA, gB, stgA, stgB, grawA, grawB); slot ^= 1; }
    if (gz + 3 <= ze) { step(gz + 3, slot, gA, gB, gC, stgB, stgA, grawB, grawA); slot ^= 1; }
    if (gz + 4 <= ze) { step(gz + 4, slot, gB, gC, gA, stgA, stgB, grawA, grawB); slot ^= 1; }
    if (gz + 5 <= ze) { step(gz + 5, slot, gC, gA, gB, stgB, stgA, grawB, grawA); slot ^= 1; }
  }

  // lanes cv, cv+16, cv+32, cv+48 hold the same channel pair: xor-shuffle sum, then waves through LDS in order
#pragma unroll
  for (int off = LPV; off < 64; off <<= 1) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[t][i] += __shfl_xor(acc[t][i], off, 64);
#pragma unroll
    for (int i = 0; i < VEC; ++i) accb[i] += __shfl_xor(accb[i], off, 64);
  }
  float* red = &plane[0][0];                       // [4 waves][28][CG]   (the last step ended with a barrier)
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < LPV) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[(wave * 28 + t) * CG + lane * VEC + i] = acc[t][i];
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(wave * 28 + 27) * CG + lane * VEC + i] = accb[i];
  }
  __syncthreads();
  const long sidx = (long)n * g.slots + slot_id;
  for (int e = tid; e < 28 * CG; e += 256) {
    const int t = e / CG, ch = e % CG;
    float a = red[(0 * 28 + t) * CG + ch];
#pragma unroll
    for (int wv = 1; wv < 4; ++wv) a += red[(wv * 28 + t) * CG + ch];
    if (t < 27) dWp[(sidx * 27 + t) * C + cg * CG + ch] = a;
    else if (dbp) dbp[sidx * C + cg * CG + ch] = a;
  }
}

static void make_wgrad_march(DwMarch& t, int N, int D, int H, int W, int C) {
  t.N = N; t.D = D; t.H = H; t.W = W; t.C = C;
  t.ty = (H + TILE_Y - 1) / TILE_Y; t.tx = (W + TILE_X - 1) / TILE_X;
  // z-chunks: ~2048 workgroups (each ends with a 3.5 KB partial), chunks of at least 14 planes
  const long fp = (long)t.ty * t.tx * (C / MARCH_CG) * N;
  int nzc = (int)((2048 + fp - 1) / fp);
  if (nzc < 1) nzc = 1;
  int maxc = D / 14;
  if (maxc < 1) maxc = 1;
  if (nzc > maxc) nzc = maxc;
  t.zc = (D + nzc - 1) / nzc;
  t.nzc = (D + t.zc - 1) / t.zc;
  t.slots = t.ty * t.tx * t.nzc;
}

// used by pytc_dw_wgrad (train_kernels.hip): slot count (0 = shape not covered) and launch of the march form
int dw_wgrad_march_slots(int N, int D, int H, int W, int C, int K, int stride, int dtype) {
  if (!march_ok(D, H, W, C, K, stride, dtype, 0) || tuning_get("dw_wgrad_march", 1) == 0) return 0;
  DwMarch t;
  make_wgrad_march(t, N, D, H, W, C);
  return t.slots * N;
}

void dw_wgrad_march_launch(const void* gr, const void* x, float* dWp, float* dbp, int N, int D, int H, int W, int C,
                           int dtype, hipStream_t s) {
  DwMarch t;
  make_wgrad_march(t, N, D, H, W, C);
  t.swizzle = tuning_get("dwconv_xcd_swizzle", 1);
  t.cg_inner = tuning_get("dwconv_cg_inner", 1);
  dim3 grid((unsigned)((long)t.slots * (C / MARCH_CG) * N)), block(256);
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(dw_wgrad_march_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)gr, (const bf16_t*)x, dWp, dbp, t);
  else
    hipLaunchKernelGGL(dw_wgrad_march_kernel<float>, grid, block, 0, s, (const float*)gr, (const float*)x, dWp, dbp, t);
}

// ---------------------------------------------------------------------------------------------
// stats [N][slots][2][C] -> ab [N][2][C]   (a = gamma*rstd, b = beta - mean*a)
// workgroup = 16 channels x 64 slot lanes (1024 threads): the slot loop is a chain of dependent-latency reads (26 launches per
// MedNeXt-S forward sit between a depthwise conv and its mixer), so it is spread over 4x the lanes and unrolled; the 4 slot
// lanes of a wave meet through shuffles, the 16 waves through LDS.
constexpr int FIN_CH = 16, FIN_SL = 64, FIN_WAVES = FIN_CH * FIN_SL / 64;
__global__ void __launch_bounds__(FIN_CH * FIN_SL)
groupnorm_finalize_kernel(const float* __restrict__ stats, int slots, float count,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                          float* __restrict__ ab, float* __restrict__ mr, int C) {
  __shared__ float red[2][FIN_WAVES][FIN_CH];
  const int n = blockIdx.y;
  const int cl = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    const float* base = stats + (long)n * slots * 2 * C;
#pragma unroll 4
    for (int s = sl; s < slots; s += FIN_SL) {
      a1 += base[((long)s * 2 + 0) * C + c];
      a2 += base[((long)s * 2 + 1) * C + c];
    }
  }
  a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
  a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
  const int wave = threadIdx.x / 64;
  if ((threadIdx.x % 64) < FIN_CH) {
    red[0][wave][cl] = a1;
    red[1][wave][cl] = a2;
  }
  __syncthreads();
  if (threadIdx.x < FIN_CH && c < C) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int s = 0; s < FIN_WAVES; ++s) { t1 += red[0][s][cl]; t2 += red[1][s][cl]; }
    float mean = t1 / count;
    float var = fmaxf(t2 / count - mean * mean, 0.f);
    float rstd = rsqrtf(var + eps);
    // rsqrtf is approximate on AMD; refine with one Newton step for fp32-grade accuracy
    rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
    float a = (gamma ? gamma[c] : 1.f) * rstd;
    float b = (beta ? beta[c] : 0.f) - mean * a;
    ab[((long)n * 2 + 0) * C + c] = a;
    ab[((long)n * 2 + 1) * C + c] = b;
    if (mr) {   // saved for the backward pass
      mr[((long)n * 2 + 0) * C + c] = mean;
      mr[((long)n * 2 + 1) * C + c] = rstd;
    }
  }
}

static int pick_vec(int C, int dtype) {
  int maxv = dtype == PYTC_BF16 ? 8 : 4;
  for (int v = maxv; v > 1; v >>= 1)
    if (C % v == 0 && (C / v) <= 256) return v;
  return C <= 256 ? 1 : 0;
}

static bool make_geom(DwGeom& g, int N, int D, int H, int W, int C, int K, int stride, int dtype,
                      int transposed, int& vec) {
  vec = pick_vec(C, dtype);
  if (!vec) return false;
  g.N = N; g.D = D; g.H = H; g.W = W; g.C = C; g.K = K; g.stride = stride;
  if (transposed) { g.Do = 2 * D; g.Ho = 2 * H; g.Wo = 2 * W; }
  else {
    int p = K / 2;
    g.Do = (D + 2 * p - K) / stride + 1; g.Ho = (H + 2 * p - K) / stride + 1; g.Wo = (W + 2 * p - K) / stride + 1;
  }
  g.lpv = C / vec;
  g.vs = 256 / g.lpv;
  g.cell = (transposed && K == 3 && 27L * C * 4 <= 64 * 1024 && tuning_get("dwconvT_cell", 1) != 0) ? 1 : 0;
  // K = 5 / 7, stride 1, 16-byte channel vectors, taps fit LDS: 4 consecutive x outputs per lane (dwconv3d_xblock_kernel)
  g.xblock = (!transposed && stride == 1 && (K == 5 || K == 7 || (K == 3 && tuning_get("dwconv_xblock_k3", 1) != 0)) && vec == 8 &&
              (long)K * K * K * C * 4 <= 64 * 1024 && tuning_get("dwconv_xblock", 1) != 0 && tuning_get("dwconv_gather", 1) != 0) ? 1 : 0;
  long vout = g.cell ? (long)D * H * W : (g.xblock ? (long)D * H * ((W + 3) / 4) : (long)g.Do * g.Ho * g.Wo);
  long it = vout / ((long)g.vs * (g.cell ? 256 : 96));   // aim for >= ~96 (cells: 256) workgroups per sample
  g.iters = (int)(it < 1 ? 1 : (it > 64 ? 64 : it));
  g.slots = (int)((vout + (long)g.vs * g.iters - 1) / ((long)g.vs * g.iters));
  return true;
}

template <typename T, int VEC>
static int launch_dw(bool transposed, const void* x, void* y, const float* w, const float* bias, float* stats,
                     const DwGeom& g, hipStream_t s) {
  dim3 grid(g.slots, g.N), block(256);
  size_t lds = stats ? (size_t)g.vs * 2 * g.C * sizeof(float) : 0;
  const size_t taps = (size_t)(transposed ? 27 : g.K * g.K * g.K) * g.C * sizeof(float);
  if (transposed && g.cell) {
    hipLaunchKernelGGL((dwconvT3d_k3_cell_kernel<T, VEC>), grid, block, lds > taps ? lds : taps, s, (const T*)x, (T*)y, w, bias, stats, g);
    return PYTC_OK;
  }
  if (!transposed && (g.K == 3 || g.K == 5 || g.K == 7) && taps <= 64 * 1024 && tuning_get("dwconv_gather", 1) != 0) {
    const size_t dyn = lds > taps ? lds : taps;
    if constexpr (VEC == 8) {
      if (g.stride == 1 && g.xblock) {
        if (g.K == 3) hipLaunchKernelGGL((dwconv3d_xblock_kernel<T, 8, 3, 4>), grid, block, dyn, s, (const T*)x, (T*)y, w, bias, stats, g);
        else if (g.K == 5) hipLaunchKernelGGL((dwconv3d_xblock_kernel<T, 8, 5, 4>), grid, block, dyn, s, (const T*)x, (T*)y, w, bias, stats, g);
        else hipLaunchKernelGGL((dwconv3d_xblock_kernel<T, 8, 7, 4>), grid, block, dyn, s, (const T*)x, (T*)y, w, bias, stats, g);
        return PYTC_OK;
      }
    }
    if (g.K == 3) hipLaunchKernelGGL((dwconv3d_k3_gather_kernel<T, VEC, 3>), grid, block, dyn, s, (const T*)x, (T*)y, w, bias, stats, g);
    else if (g.K == 5) hipLaunchKernelGGL((dwconv3d_k3_gather_kernel<T, VEC, 5>), grid, block, dyn, s, (const T*)x, (T*)y, w, bias, stats, g);
    else hipLaunchKernelGGL((dwconv3d_k3_gather_kernel<T, VEC, 7>), grid, block, dyn, s, (const T*)x, (T*)y, w, bias, stats, g);
    return PYTC_OK;
  }
#define PYTC_DW_CASE(KK)                                                                                      \
  case KK:                                                                                                    \
    if (transposed)                                                                                           \
      hipLaunchKernelGGL((dwconvT3d_kernel<T, VEC, KK>), grid, block, lds, s, (const T*)x, (T*)y, w, bias, stats, g); \
    else                                                                                                      \
      hipLaunchKernelGGL((dwconv3d_direct_kernel<T, VEC, KK>), grid, block, lds, s, (const T*)x, (T*)y, w, bias, stats, g); \
    break;
  switch (g.K) {
    PYTC_DW_CASE(1)
    PYTC_DW_CASE(3)
    PYTC_DW_CASE(5)
    PYTC_DW_CASE(7)
    default:
      set_error("dwconv3d: unsupported kernel size %d", g.K);
      return PYTC_ERR_UNSUPPORTED;
  }
#undef PYTC_DW_CASE
  return PYTC_OK;
}

template <typename T>
static int dispatch_vec(int vec, bool transposed, const void* x, void* y, const float* w, const float* bias,
                        float* stats, const DwGeom& g, hipStream_t s) {
  switch (vec) {
    case 8: if constexpr (sizeof(T) == 2) return launch_dw<T, 8>(transposed, x, y, w, bias, stats, g, s); else break;
    case 4: return launch_dw<T, 4>(transposed, x, y, w, bias, stats, g, s);
    case 2: return launch_dw<T, 2>(transposed, x, y, w, bias, stats, g, s);
    case 1: return launch_dw<T, 1>(transposed, x, y, w, bias, stats, g, s);
  }
  set_error("dwconv3d: bad vector width %d", vec);
  return PYTC_ERR_INVALID;
}

static int dw_entry(bool transposed, const void* x, void* y, const float* w, const float* bias, float* stats,
                    int N, int D, int H, int W, int C, int K, int stride, int dtype, void* stream,
                    const void* res = nullptr, bool wide_range = false) {
  PYTC_REQUIRE(x && w, "dwconv3d: null pointer");
  if (res) {      // y = conv(x) + res: the z-march kernel only (stride 1, K = 3, C % 32 == 0, planes >= 16 x 16, depth >= 8)
    PYTC_REQUIRE(y && !stats, "dwconv3d_res: needs an output and takes no statistics");
    if (dtype != PYTC_BF16 || !march_ok(D, H, W, C, K, stride, dtype, transposed)) {
      set_error("dwconv3d_res: bf16 only, shape outside the z-march kernel (N=%d D=%d H=%d W=%d C=%d K=%d stride=%d)", N, D, H, W, C, K, stride);
      return PYTC_ERR_UNSUPPORTED;
    }
  }
  PYTC_REQUIRE(N >= 1 && D >= 1 && H >= 1 && W >= 1 && C >= 1, "dwconv3d: bad shape");
  PYTC_REQUIRE(stride == 1 || stride == 2, "dwconv3d: stride must be 1 or 2");
  PYTC_REQUIRE(dtype == PYTC_F32 || dtype == PYTC_BF16, "dwconv3d: bad dtype");
  PYTC_REQUIRE(!(wide_range && stats), "dwconv3d_fwd_wide: the gradient entry computes no statistics");
  // the up blocks' resampling conv at C = 64 / 128: one tile of input cells per workgroup (dwconvT_tile_kernels.hip).  The plan is made
  // FIRST: a shape it rejects falls through to the generic kernels, which take a null output only in their cell form (checked below)
  DwTTile tt;
  const bool tt_form = transposed && K == 3 && dtype == PYTC_BF16 && (C == 64 || C == 128) && tuning_get("dwconvT_tile", 1) != 0 &&
                       dwconvT_tile_plan(tt, N, D, H, W, C);
  // statistics-only passes (null output): the matrix-core stride-1 kernel (the fused block's first pass, pw_dwmix_kernels.hip) and
  // the K = 3 transposed kernels (tile form / cell form: the fused up-block path)
  const bool mfma_form = !transposed && !res && !wide_range && dtype == PYTC_BF16 && tuning_get("dwconv_mfma", 1) != 0 &&
                         (march_ok(D, H, W, C, K, stride, dtype, transposed) || mfma_small_ok(D, H, W, C, K, stride, dtype, transposed));
  if (!y) {
    PYTC_REQUIRE(stats, "dwconv3d: neither output nor statistics requested");
    if (!tt_form && !mfma_form) {
      DwGeom g0;
      int vec0;
      PYTC_REQUIRE(transposed && K == 3 && make_geom(g0, N, D, H, W, C, K, stride, dtype, transposed, vec0) && g0.cell,
                   "dwconv3d: a null output (statistics only) is supported by the bf16 matrix-core stride-1 kernel and the K = 3 "
                   "transposed kernels only");
    }
  }
  if (tt_form) {
    dwconvT_tile_launch(x, y, w, bias, stats, tt, (hipStream_t)stream);
    PYTC_LAUNCH_CHECK("dwconvT3d_k3_tile");
    return PYTC_OK;
  }
  if (!transposed && K == 3 && stride == 2 && dtype == PYTC_BF16 && y && tuning_get("dwconv_s2_march", 1) != 0) {
    DwS2 t2;          // the down blocks' resampling conv at C = 32 / 64: z-march over an LDS ring (dwconv_s2_kernels.hip)
    if (dwconv_s2_plan(t2, N, D, H, W, C)) {
      dwconv_s2_launch(x, y, w, bias, stats, t2, (hipStream_t)stream);
      PYTC_LAUNCH_CHECK("dwconv3d_k3_s2_march");
      return PYTC_OK;
    }
  }
  if (!res && !wide_range && mfma_small_ok(D, H, W, C, K, stride, dtype, transposed)) {
    DwMarch t;
    make_march(t, N, D, H, W, C, TILE_X);
    t.swizzle = tuning_get("dwconv_xcd_swizzle", 1);
    t.cg_inner = tuning_get("dwconv_cg_inner", 1);
    // hi + lo weights here: these launches are latency bound (22 us either way), and with bf16 weights at the 14^3 level the
    // training gate's worst tensor (bottleneck.1.norm.weight) moved from 0.050 to 0.066 relative L2 against a 0.05 class
    dwconv_mfma_launch(x, y, w, bias, stats, t, tuning_get("dwconv_mfma_variant", 0) | 1, (hipStream_t)stream);
    PYTC_LAUNCH_CHECK("dwconv3d_k3_mfma");
    return PYTC_OK;
  }
  if (march_ok(D, H, W, C, K, stride, dtype, transposed)) {
    DwMarch t;
    // the 8 x 16 / 512-thread footprint serves the plain packed-f16 forward (the launches that carry statistics, so the slot
    // count of pytc_dwconv3d_stat_slots follows the same rule); gradient entries keep the 8 x 8 kernels
    const bool wide_tile = !res && !wide_range && tuning_get("dwconv_march_h16", 1) != 0 && march_tile_x(W, dtype) == 16;
    make_march(t, N, D, H, W, C, wide_tile ? 16 : TILE_X);
    dim3 grid((unsigned)((long)t.slots * (C / MARCH_CG) * N)), block(wide_tile ? 512 : 256);
    t.swizzle = tuning_get("dwconv_xcd_swizzle", 1);
  t.cg_inner = tuning_get("dwconv_cg_inner", 1);
    // bf16 forward launches (activations: statistics, no residual): the matrix-core form (dwconv_mfma_kernels.hip) -- one channel per
    // block of v_mfma_f32_4x4x4_16b_bf16, fp32 accumulation.  The gradient entries (res / wide range) keep the fp32-tap VALU kernels.
    if (dtype == PYTC_BF16 && !res && !wide_range && tuning_get("dwconv_mfma", 1) != 0) {
      dwconv_mfma_launch(x, y, w, bias, stats, t, tuning_get("dwconv_mfma_variant", 0), (hipStream_t)stream);
      PYTC_LAUNCH_CHECK("dwconv3d_k3_mfma");
      return PYTC_OK;
    }
    // variants (all: taps in LDS, hand-scheduled tap loop, asm plane loads with counted waits):
    //   0 (default) PF=3 compiled for 4 waves/SIMD; 1: PF=3, 3 waves; 2: PF=3, 2 waves; 3: PF=2, 3 waves
    const int variant = tuning_get("dwconv_march_variant", 0);
#define PYTC_MARCH(PP, WW) \
  hipLaunchKernelGGL((dwconv3d_k3_march_kernel<bf16_t, 2, PP, true, WW>), grid, block, 0, (hipStream_t)stream, \
                     (const bf16_t*)x, (bf16_t*)y, w, bias, stats, t)
    // packed-f16 in-plane partial sums (see the kernel header); 0: fp32 taps.  f16 resolves 6e-8 at best: it is for ACTIVATIONS.
    // Gradient operands (a mean-reduced loss over 1.4 M voxels gives |dL/dx| ~ 1e-7) sit in its subnormal range -- measured:
    // every parameter gradient behind the last block off by 35-85 % at 112^3 (tests/test_gpu_baseline_sizes.py training gate,
    // profiles/r03_training_gradient_gate.txt) -- so the residual / wide-range entries always take the fp32-tap kernel
    const int h16 = (res || wide_range) ? 0 : tuning_get("dwconv_march_h16", 1);
    if (res) {
      // 8 more live registers than the plain kernel (two residual sets in flight): compiled for 3 waves / SIMD.  At the
      // 4-waves budget (128 VGPRs) hipcc spills 31 registers, and a spill of a register an asm-issued load is still
      // writing corrupts the value -- the counted-wait scheme REQUIRES a spill-free kernel (measured: garbage output)
      if (h16)
        hipLaunchKernelGGL((dwconv3d_k3_march_kernel<bf16_t, 2, 3, true, 3, true, true>), grid, block, 0, (hipStream_t)stream,
                           (const bf16_t*)x, (bf16_t*)y, w, bias, stats, t, (const bf16_t*)res);
      else
        hipLaunchKernelGGL((dwconv3d_k3_march_kernel<bf16_t, 2, 3, true, 3, true>), grid, block, 0, (hipStream_t)stream,
                           (const bf16_t*)x, (bf16_t*)y, w, bias, stats, t, (const bf16_t*)res);
    } else if (dtype == PYTC_BF16 && h16 && wide_tile) {
      hipLaunchKernelGGL((dwconv3d_k3_march_kernel<bf16_t, 2, 3, true, 2, false, true, 16, 512>), grid, block, 0, (hipStream_t)stream,
                         (const bf16_t*)x, (bf16_t*)y, w, bias, stats, t);
    } else if (dtype == PYTC_BF16 && h16) {
      hipLaunchKernelGGL((dwconv3d_k3_march_kernel<bf16_t, 2, 3, true, 4, false, true>), grid, block, 0, (hipStream_t)stream,
                         (const bf16_t*)x, (bf16_t*)y, w, bias, stats, t);
    } else if (dtype == PYTC_BF16) {
      switch (variant) {
        case 1: PYTC_MARCH(3, 3); break;
        case 2: PYTC_MARCH(3, 2); break;
        case 3: PYTC_MARCH(2, 3); break;
        default: PYTC_MARCH(3, 4); break;
      }
    } else {
      hipLaunchKernelGGL((dwconv3d_k3_march_kernel<float, 2, 1, false, 2>), grid, block, 0, (hipStream_t)stream,
                         (const float*)x, (float*)y, w, bias, stats, t);
    }
#undef PYTC_MARCH
    PYTC_LAUNCH_CHECK("dwconv3d_k3_march");
    return PYTC_OK;
  }
  DwGeom g;
  int vec;
  if (!make_geom(g, N, D, H, W, C, K, stride, dtype, transposed, vec)) {
    set_error("dwconv3d: unsupported channel count %d", C);
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_REQUIRE((size_t)g.vs * 2 * C * sizeof(float) <= 64 * 1024, "dwconv3d: stats scratch too large");
  int rc = dtype == PYTC_BF16 ? dispatch_vec<bf16_t>(vec, transposed, x, y, w, bias, stats, g, (hipStream_t)stream)
                              : dispatch_vec<float>(vec, transposed, x, y, w, bias, stats, g, (hipStream_t)stream);
  if (rc != PYTC_OK) return rc;
  PYTC_LAUNCH_CHECK("dwconv3d");
  return PYTC_OK;
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_dwconv3d_stat_slots(int N, int D, int H, int W, int C, int K, int stride, int dtype,
                                        int transposed) {
  if (march_ok(D, H, W, C, K, stride, dtype, transposed)) {
    DwMarch t;
    make_march(t, N, D, H, W, C, tuning_get("dwconv_march_h16", 1) != 0 ? march_tile_x(W, dtype) : TILE_X);
    return t.slots;
  }
  if (mfma_small_ok(D, H, W, C, K, stride, dtype, transposed)) {
    DwMarch t;
    make_march(t, N, D, H, W, C, TILE_X);
    return t.slots;
  }
  if (transposed && K == 3 && dtype == PYTC_BF16 && (C == 64 || C == 128) && tuning_get("dwconvT_tile", 1) != 0) {
    DwTTile tt;
    if (dwconvT_tile_plan(tt, N, D, H, W, C)) return tt.slots;
  }
  if (!transposed && K == 3 && stride == 2 && dtype == PYTC_BF16 && tuning_get("dwconv_s2_march", 1) != 0) {
    DwS2 t2;
    if (dwconv_s2_plan(t2, N, D, H, W, C)) return t2.slots;
  }
  DwGeom g;
  int vec;
  if (!make_geom(g, N, D, H, W, C, K, stride, dtype, transposed, vec)) return -1;
  return g.slots;
}

extern "C" int pytc_dwconv3d_kernel_variant(int N, int D, int H, int W, int C, int K, int stride, int dtype, int transposed) {
  // mirrors dw_entry / launch_dw: which kernel family a call with these arguments dispatches to
  if (march_ok(D, H, W, C, K, stride, dtype, transposed)) return (dtype == PYTC_BF16 && tuning_get("dwconv_mfma", 1) != 0) ? 6 : 3;
  if (mfma_small_ok(D, H, W, C, K, stride, dtype, transposed)) return 6;
  DwGeom g;
  int vec;
  if (!make_geom(g, N, D, H, W, C, K, stride, dtype, transposed, vec)) return -1;
  if (transposed && K == 3 && dtype == PYTC_BF16 && (C == 64 || C == 128) && tuning_get("dwconvT_tile", 1) != 0) return 7;
  if (!transposed && K == 3 && stride == 2 && dtype == PYTC_BF16 && (C == 32 || C == 64) && tuning_get("dwconv_s2_march", 1) != 0) return 8;
  if (transposed) return g.cell ? 4 : 5;
  const size_t taps = (size_t)K * K * K * C * sizeof(float);
  if ((K == 3 || K == 5 || K == 7) && taps <= 64 * 1024 && tuning_get("dwconv_gather", 1) != 0)
    return (vec == 8 && stride == 1 && g.xblock) ? 2 : 1;
  return 0;
}

extern "C" int pytc_dwconv3d_fwd(const void* x, void* y, const float* w, const float* bias, float* stats, int N,
                                 int D, int H, int W, int C, int K, int stride, int dtype, void* stream) {
  return dw_entry(false, x, y, w, bias, stats, N, D, H, W, C, K, stride, dtype, stream);
}

extern "C" int pytc_dwconv3d_fwd_wide(const void* x, void* y, const float* w, const float* bias, float* stats, int N,
                                      int D, int H, int W, int C, int K, int stride, int dtype, void* stream) {
  return dw_entry(false, x, y, w, bias, stats, N, D, H, W, C, K, stride, dtype, stream, nullptr, true);
}

extern "C" int pytc_dwconv3d_res_supported(int D, int H, int W, int C, int K, int stride, int dtype) {
  return (dtype == PYTC_BF16 && march_ok(D, H, W, C, K, stride, dtype, 0)) ? 1 : 0;
}

extern "C" int pytc_dwconv3d_fwd_res(const void* x, const void* res, void* y, const float* w, const float* bias, int N,
                                     int D, int H, int W, int C, int K, int stride, int dtype, void* stream) {
  PYTC_REQUIRE(res, "dwconv3d_res: null residual");
  return dw_entry(false, x, y, w, bias, nullptr, N, D, H, W, C, K, stride, dtype, stream, res);
}

extern "C" int pytc_dwmix_supported(int D, int H, int W, int C, int C_hid, int C_out, int dtype) {
  return (dtype == PYTC_BF16 && C == 32 && C_out == 32 && (C_hid == 64 || C_hid == 96 || C_hid == 128) && tuning_get("dwconv_mfma", 1) != 0 &&
          (march_ok(D, H, W, C, 3, 1, dtype, 0) || mfma_small_ok(D, H, W, C, 3, 1, dtype, 0))) ? 1 : 0;
}

extern "C" int pytc_dwmix_fwd(const void* x, const float* taps, const float* dw_bias, const void* w2n, const float* b2n,
                              const void* w3_f16, const float* b3, int residual, void* y, const void* head_w, const float* head_b,
                              float* head_y, int n_head, int N, int D, int H, int W, int C, int C_hid, int C_out, int dtype,
                              void* stream) {
  PYTC_REQUIRE(x && taps && w2n && b2n && w3_f16 && b3, "dwmix: null pointer");
  PYTC_REQUIRE(N >= 1 && pytc_dwmix_supported(D, H, W, C, C_hid, C_out, dtype),
               "dwmix: bf16, C = C_out = 32, C_hid in {64, 96, 128}, a shape of the matrix-core depthwise kernel (D >= 8, H, W >= 8)");
  PYTC_REQUIRE(y || (head_w && head_y), "dwmix: neither the block output nor the head output requested");
  PYTC_REQUIRE(!head_w || (head_y && n_head >= 1 && n_head <= 16), "dwmix: the fused head writes 1..16 channels to head_y");
  DwMarch t;
  make_march(t, N, D, H, W, C, TILE_X);
  t.swizzle = tuning_get("dwconv_xcd_swizzle", 1);
  t.cg_inner = tuning_get("dwconv_cg_inner", 1);
  DwMix mx{};
  mx.w2n = (const bf16x8_t*)w2n; mx.b2n = b2n; mx.w3 = (const h8_t*)w3_f16; mx.b3 = b3;
  mx.w2_stride = (long)C_hid * C / 8;
  mx.residual = residual ? 1 : 0;
  mx.head_w = (const bf16x8_t*)head_w; mx.head_b = head_b; mx.head_y = head_y; mx.n_head = n_head; mx.store_y = y ? 1 : 0;
  // measurement only (knob dwconv_mfma_probe = 4): the address of a [N][slots][4][9] fp32 buffer in the knobs dwmix_prof_lo / _hi
  if (tuning_get("dwconv_mfma_probe", 0) == 4)
    mx.prof = (float*)(((unsigned long long)(unsigned)tuning_get("dwmix_prof_hi", 0) << 32) | (unsigned)tuning_get("dwmix_prof_lo", 0));
  // the statistics this block was normalised with are those of pytc_dwconv3d_fwd(y = NULL) under the same knobs: same variant here
  // (small planes -- the 8 <= H, W < 16 launches -- always carry hi + lo weights there)
  int variant = tuning_get("dwconv_mfma_variant", 0);
  if (!march_ok(D, H, W, C, 3, 1, dtype, 0)) variant |= 1;
  PYTC_REQUIRE(dwmix_launch(x, y, taps, dw_bias, t, mx, C_hid, variant, (hipStream_t)stream) == 0, "dwmix: unsupported hidden width");
  PYTC_LAUNCH_CHECK("dwmix");
  return PYTC_OK;
}

extern "C" int pytc_dwconvT3d_fwd(const void* x, void* y, const float* w, const float* bias, float* stats, int N,
                                  int D, int H, int W, int C, int K, int dtype, void* stream) {
  return dw_entry(true, x, y, w, bias, stats, N, D, H, W, C, K, 2, dtype, stream);
}

extern "C" int pytc_groupnorm_finalize_mr(const float* stats, int slots, float count, const float* gamma,
                                          const float* beta, float eps, float* ab, float* mean_rstd, int N, int C,
                                          void* stream) {
  PYTC_REQUIRE(stats && ab && slots >= 1 && count > 0 && N >= 1 && C >= 1, "groupnorm_finalize: bad arguments");
  dim3 grid(ceil_div(C, FIN_CH), N), block(FIN_CH * FIN_SL);
  hipLaunchKernelGGL(groupnorm_finalize_kernel, grid, block, 0, (hipStream_t)stream, stats, slots, count, gamma, beta,
                     eps, ab, mean_rstd, C);
  PYTC_LAUNCH_CHECK("groupnorm_finalize");
  return PYTC_OK;
}

extern "C" int pytc_groupnorm_finalize(const float* stats, int slots, float count, const float* gamma,
                                       const float* beta, float eps, float* ab, int N, int C, void* stream) {
  return pytc_groupnorm_finalize_mr(stats, slots, count, gamma, beta, eps, ab, nullptr, N, C, stream);
}
