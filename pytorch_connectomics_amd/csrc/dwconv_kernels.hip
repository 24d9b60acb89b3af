// Depthwise 3-D convolution kernels (NDHWC, lanes along C so every global access is a
// coalesced vector of VEC channels), fused with the per-(n,c) sum / sum-of-squares the
// following GroupNorm(C, C) needs.  Statistics are written as per-workgroup partials and
// reduced in a fixed order by groupnorm_finalize -> bit-reproducible, no float atomics.
#include "pytc_common.h"

namespace pytc {

struct DwGeom {
  int N, D, H, W, C, K, stride, Do, Ho, Wo;
  int lpv;        // lanes per voxel = C / VEC
  int vs;         // voxel slots per pass = 256 / lpv
  int iters;      // passes per workgroup
  int slots;      // workgroups per sample
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

// ---------------------------------------------------------------------------------------------
// stats [N][slots][2][C] -> ab [N][2][C]   (a = gamma*rstd, b = beta - mean*a)
constexpr int FIN_CH = 16, FIN_SL = 16;
__global__ void __launch_bounds__(256)
groupnorm_finalize_kernel(const float* __restrict__ stats, int slots, float count,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                          float* __restrict__ ab, int C) {
  __shared__ float red[2][FIN_SL][FIN_CH];
  const int n = blockIdx.y;
  const int cl = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    const float* base = stats + (long)n * slots * 2 * C;
    for (int s = sl; s < slots; s += FIN_SL) {
      a1 += base[((long)s * 2 + 0) * C + c];
      a2 += base[((long)s * 2 + 1) * C + c];
    }
  }
  red[0][sl][cl] = a1;
  red[1][sl][cl] = a2;
  __syncthreads();
  if (sl == 0 && c < C) {
    float t1 = 0.f, t2 = 0.f;
    for (int s = 0; s < FIN_SL; ++s) { t1 += red[0][s][cl]; t2 += red[1][s][cl]; }
    float mean = t1 / count;
    float var = fmaxf(t2 / count - mean * mean, 0.f);
    float rstd = rsqrtf(var + eps);
    // rsqrtf is approximate on AMD; refine with one Newton step for fp32-grade accuracy
    rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
    float a = (gamma ? gamma[c] : 1.f) * rstd;
    float b = (beta ? beta[c] : 0.f) - mean * a;
    ab[((long)n * 2 + 0) * C + c] = a;
    ab[((long)n * 2 + 1) * C + c] = b;
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
  long vout = (long)g.Do * g.Ho * g.Wo;
  long it = vout / ((long)g.vs * 96);   // aim for >= ~96 workgroups per sample
  g.iters = (int)(it < 1 ? 1 : (it > 64 ? 64 : it));
  g.slots = (int)((vout + (long)g.vs * g.iters - 1) / ((long)g.vs * g.iters));
  return true;
}

template <typename T, int VEC>
static int launch_dw(bool transposed, const void* x, void* y, const float* w, const float* bias, float* stats,
                     const DwGeom& g, hipStream_t s) {
  dim3 grid(g.slots, g.N), block(256);
  size_t lds = stats ? (size_t)g.vs * 2 * g.C * sizeof(float) : 0;
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
                    int N, int D, int H, int W, int C, int K, int stride, int dtype, void* stream) {
  PYTC_REQUIRE(x && y && w, "dwconv3d: null pointer");
  PYTC_REQUIRE(N >= 1 && D >= 1 && H >= 1 && W >= 1 && C >= 1, "dwconv3d: bad shape");
  PYTC_REQUIRE(stride == 1 || stride == 2, "dwconv3d: stride must be 1 or 2");
  PYTC_REQUIRE(dtype == PYTC_F32 || dtype == PYTC_BF16, "dwconv3d: bad dtype");
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

extern "C" int pytc_dwconv3d_stat_slots(int D, int H, int W, int C, int K, int stride, int dtype, int transposed) {
  DwGeom g;
  int vec;
  if (!make_geom(g, 1, D, H, W, C, K, stride, dtype, transposed, vec)) return -1;
  return g.slots;
}

extern "C" int pytc_dwconv3d_fwd(const void* x, void* y, const float* w, const float* bias, float* stats, int N,
                                 int D, int H, int W, int C, int K, int stride, int dtype, void* stream) {
  return dw_entry(false, x, y, w, bias, stats, N, D, H, W, C, K, stride, dtype, stream);
}

extern "C" int pytc_dwconvT3d_fwd(const void* x, void* y, const float* w, const float* bias, float* stats, int N,
                                  int D, int H, int W, int C, int K, int dtype, void* stream) {
  return dw_entry(true, x, y, w, bias, stats, N, D, H, W, C, K, 2, dtype, stream);
}

extern "C" int pytc_groupnorm_finalize(const float* stats, int slots, float count, const float* gamma,
                                       const float* beta, float eps, float* ab, int N, int C, void* stream) {
  PYTC_REQUIRE(stats && ab && slots >= 1 && count > 0 && N >= 1 && C >= 1, "groupnorm_finalize: bad arguments");
  dim3 grid(ceil_div(C, FIN_CH), N), block(256);
  hipLaunchKernelGGL(groupnorm_finalize_kernel, grid, block, 0, (hipStream_t)stream, stats, slots, count, gamma, beta,
                     eps, ab, C);
  PYTC_LAUNCH_CHECK("groupnorm_finalize");
  return PYTC_OK;
}
