// Disk-backed test volumes on the device (reference connectomics/inference/lazy.py:456-917,
// data/augmentation/augment_ops.py:552-611).
//
// The reference's LazyVolumeAccessor transposes, resizes, context-pads and normalises every window in numpy on the host.
// Here the host only reads the RAW bytes of the storage box a region needs (storage dtype, storage axis order) into pinned
// memory; everything else is two kernels:
//
//   resample_region  : raw box (any integer / float storage type, any axis order through element strides) -> fp32 (C, z, y, x)
//                      of the transformed, context-padded volume.  Per output axis the host supplies a small table
//                      (i0, i1, f): the two raw indices an output index reads and the weight of the second one -- which
//                      encodes transpose (via the strides), nearest / trilinear resize and constant / reflect / edge context
//                      padding at once (i0 < 0: outside a constant pad -> 0).
//   window_normalize : the per-WINDOW part of the pipeline after the window gather (outer padding included, as in
//                      read_patch): optional binarisation, percentile clip bounds, then z-score / min-max / divide-K, with
//                      the window's statistics from a fixed-order two-stage reduction in fp64.
#include "pytc_common.h"

namespace pytc {

template <typename T>
__device__ __forceinline__ float raw_to_f32(T v) { return (float)v; }

struct AxisTab {
  const int32_t* i0;
  const int32_t* i1;
  const float* f;
  int n;
};

// one thread per output voxel (x fastest), channels in a loop; 8 taps, fewer when an axis is not interpolated (f == 0)
template <typename T>
__global__ void __launch_bounds__(256)
resample_region_kernel(const T* __restrict__ raw, long sc, long sz, long sy, long sx, int C, AxisTab tz, AxisTab ty, AxisTab tx,
                       float* __restrict__ out) {
  const long total = (long)tz.n * ty.n * tx.n;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % tx.n);
  const long t = i / tx.n;
  const int y = (int)(t % ty.n), z = (int)(t / ty.n);
  const int z0 = tz.i0[z], y0 = ty.i0[y], x0 = tx.i0[x];
  if (z0 < 0 || y0 < 0 || x0 < 0) {                      // outside a constant context pad
    for (int c = 0; c < C; ++c) out[(long)c * total + i] = 0.f;
    return;
  }
  const int z1 = tz.i1[z], y1 = ty.i1[y], x1 = tx.i1[x];
  const float fz = tz.f[z], fy = ty.f[y], fx = tx.f[x];
  const long a00 = z0 * sz + y0 * sy, a01 = z0 * sz + y1 * sy, a10 = z1 * sz + y0 * sy, a11 = z1 * sz + y1 * sy;
  const long b0 = x0 * sx, b1 = x1 * sx;
  const bool lin = fz != 0.f || fy != 0.f || fx != 0.f;
  for (int c = 0; c < C; ++c) {
    const T* p = raw + c * sc;
    float v;
    if (!lin) {
      v = raw_to_f32(p[a00 + b0]);
    } else {
      // separable trilinear blend; weights (1 - f, f) per axis as torch's grid_sample forms them (align_corners=True)
      const float gx0 = 1.f - fx, gy0 = 1.f - fy, gz0 = 1.f - fz;
      const float r00 = __fadd_rn(__fmul_rn(raw_to_f32(p[a00 + b0]), gx0), __fmul_rn(raw_to_f32(p[a00 + b1]), fx));
      const float r01 = __fadd_rn(__fmul_rn(raw_to_f32(p[a01 + b0]), gx0), __fmul_rn(raw_to_f32(p[a01 + b1]), fx));
      const float r10 = __fadd_rn(__fmul_rn(raw_to_f32(p[a10 + b0]), gx0), __fmul_rn(raw_to_f32(p[a10 + b1]), fx));
      const float r11 = __fadd_rn(__fmul_rn(raw_to_f32(p[a11 + b0]), gx0), __fmul_rn(raw_to_f32(p[a11 + b1]), fx));
      const float q0 = __fadd_rn(__fmul_rn(r00, gy0), __fmul_rn(r01, fy));
      const float q1 = __fadd_rn(__fmul_rn(r10, gy0), __fmul_rn(r11, fy));
      v = __fadd_rn(__fmul_rn(q0, gz0), __fmul_rn(q1, fz));
    }
    out[(long)c * total + i] = v;
  }
}

// ---- per-window statistics: partial (sum, sum of squares, min, max) of the binarised / clipped values per (window, slot) --------
constexpr int WS_ITEMS = 8192;      // elements per workgroup of the statistics pass

__device__ __forceinline__ float prep_value(float v, int binarize, float thr, const float* clip, int b) {
  if (binarize) v = v > thr ? 1.f : 0.f;
  if (clip) v = fminf(fmaxf(v, clip[2 * b]), clip[2 * b + 1]);
  return v;
}

__global__ void __launch_bounds__(256)
window_stats_kernel(const float* __restrict__ x, long n, int slots, int binarize, float thr, const float* __restrict__ clip,
                    double* __restrict__ part) {
  const int b = blockIdx.y, slot = blockIdx.x;
  const float* xb = x + (long)b * n;
  const long lo = (long)slot * WS_ITEMS, hi = lo + WS_ITEMS < n ? lo + WS_ITEMS : n;
  double s = 0.0, q = 0.0;
  float mn = INFINITY, mx = -INFINITY;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = prep_value(xb[i], binarize, thr, clip, b);
    s += (double)v;
    q += (double)v * (double)v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  __shared__ double rs[256], rq[256];
  __shared__ float rmn[256], rmx[256];
  rs[threadIdx.x] = s; rq[threadIdx.x] = q; rmn[threadIdx.x] = mn; rmx[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {                     // fixed tree: bit-reproducible
    if ((int)threadIdx.x < w) {
      rs[threadIdx.x] += rs[threadIdx.x + w];
      rq[threadIdx.x] += rq[threadIdx.x + w];
      rmn[threadIdx.x] = fminf(rmn[threadIdx.x], rmn[threadIdx.x + w]);
      rmx[threadIdx.x] = fmaxf(rmx[threadIdx.x], rmx[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* o = part + ((long)b * slots + slot) * 4;
    o[0] = rs[0]; o[1] = rq[0]; o[2] = (double)rmn[0]; o[3] = (double)rmx[0];
  }
}

// per window: (shift, scale) such that out = (v - shift) * scale;  z-score: (mean, 1/std) when std > 1e-8; min-max: (min, 1/(max-min))
// when max > min; identity otherwise (augment_ops.py:585-597)
__global__ void __launch_bounds__(64)
window_stats_finalize_kernel(const double* __restrict__ part, int slots, long n, int mode, float* __restrict__ coef) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  double s = 0.0, q = 0.0, mn = INFINITY, mx = -INFINITY;
  for (int k = 0; k < slots; ++k) {                       // fixed order
    const double* p = part + ((long)b * slots + k) * 4;
    s += p[0]; q += p[1];
    mn = p[2] < mn ? p[2] : mn;
    mx = p[3] > mx ? p[3] : mx;
  }
  float shift = 0.f, scale = 1.f;
  if (mode == PYTC_NORM_ZSCORE) {
    const double mean = s / (double)n;
    double var = q / (double)n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double sd = sqrt(var);
    if (sd > 1e-8) { shift = (float)mean; scale = (float)(1.0 / sd); }
  } else if (mode == PYTC_NORM_MINMAX) {
    if (mx > mn) { shift = (float)mn; scale = (float)(1.0 / (mx - mn)); }
  }
  coef[2 * b] = shift;
  coef[2 * b + 1] = scale;
}

__global__ void __launch_bounds__(256)
window_apply_kernel(float* __restrict__ x, long n, int B, int binarize, float thr, const float* __restrict__ clip,
                    const float* __restrict__ coef, int mode, float divide) {
  const long total = (long)B * n;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int b = (int)(i / n);
    float v = prep_value(x[i], binarize, thr, clip, b);
    if (mode == PYTC_NORM_DIVIDE) v = __fdiv_rn(v, divide);
    else if (mode == PYTC_NORM_ZSCORE) v = __fmul_rn(__fsub_rn(v, coef[2 * b]), coef[2 * b + 1]);
    else if (mode == PYTC_NORM_MINMAX) v = __fmul_rn(__fsub_rn(v, coef[2 * b]), coef[2 * b + 1]);
    x[i] = v;
  }
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_resample_region(const void* raw, int raw_dtype, const int64_t* strides_czyx, int C, const int32_t* tab_i0,
                                    const int32_t* tab_i1, const float* tab_f, const int32_t* dims_zyx, float* out, void* stream) {
  PYTC_REQUIRE(raw && strides_czyx && tab_i0 && tab_i1 && tab_f && dims_zyx && out, "resample_region: null pointer");
  const int nz = dims_zyx[0], ny = dims_zyx[1], nx = dims_zyx[2];
  PYTC_REQUIRE(C >= 1 && nz >= 1 && ny >= 1 && nx >= 1, "resample_region: bad shape");
  // the three tables are stored back to back: [z | y | x]
  AxisTab tz{tab_i0, tab_i1, tab_f, nz}, ty{tab_i0 + nz, tab_i1 + nz, tab_f + nz, ny},
      tx{tab_i0 + nz + ny, tab_i1 + nz + ny, tab_f + nz + ny, nx};
  const long total = (long)nz * ny * nx;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const long sc = strides_czyx[0], sz = strides_czyx[1], sy = strides_czyx[2], sx = strides_czyx[3];
#define PYTC_RS(T) hipLaunchKernelGGL(resample_region_kernel<T>, grid, block, 0, s, (const T*)raw, sc, sz, sy, sx, C, tz, ty, tx, out)
  switch (raw_dtype) {
    case PYTC_RAW_U8: PYTC_RS(uint8_t); break;
    case PYTC_RAW_I8: PYTC_RS(int8_t); break;
    case PYTC_RAW_U16: PYTC_RS(uint16_t); break;
    case PYTC_RAW_I16: PYTC_RS(int16_t); break;
    case PYTC_RAW_U32: PYTC_RS(uint32_t); break;
    case PYTC_RAW_I32: PYTC_RS(int32_t); break;
    case PYTC_RAW_F32: PYTC_RS(float); break;
    case PYTC_RAW_F64: PYTC_RS(double); break;
    default: PYTC_REQUIRE(false, "resample_region: unsupported storage dtype code %d", raw_dtype);
  }
#undef PYTC_RS
  PYTC_LAUNCH_CHECK("resample_region");
  return PYTC_OK;
}

extern "C" int64_t pytc_window_normalize_ws_elems(int B, int64_t n) {
  if (B < 1 || n < 1) return -1;
  const int64_t slots = (n + WS_ITEMS - 1) / WS_ITEMS;
  return (int64_t)B * slots * 4 + (int64_t)B;      // doubles: partials, then (shift, scale) as B x 2 floats = B doubles
}

extern "C" int pytc_window_normalize(float* x, int B, int64_t n, int mode, int binarize, float threshold, float divide,
                                     const float* clip, double* workspace, void* stream) {
  PYTC_REQUIRE(x && B >= 1 && n >= 1, "window_normalize: bad arguments");
  PYTC_REQUIRE(mode == PYTC_NORM_NONE || mode == PYTC_NORM_ZSCORE || mode == PYTC_NORM_MINMAX || mode == PYTC_NORM_DIVIDE,
               "window_normalize: bad mode %d", mode);
  PYTC_REQUIRE(mode != PYTC_NORM_DIVIDE || divide != 0.f, "window_normalize: 'divide' needs a non-zero divisor");
  hipStream_t s = (hipStream_t)stream;
  float* coef = nullptr;
  if (mode == PYTC_NORM_ZSCORE || mode == PYTC_NORM_MINMAX) {
    PYTC_REQUIRE(workspace, "window_normalize: statistics modes need the workspace (pytc_window_normalize_ws_elems doubles)");
    const int slots = (int)((n + WS_ITEMS - 1) / WS_ITEMS);
    coef = reinterpret_cast<float*>(workspace + (long)B * slots * 4);
    hipLaunchKernelGGL(window_stats_kernel, dim3(slots, B), dim3(256), 0, s, x, (long)n, slots, binarize, threshold, clip, workspace);
    hipLaunchKernelGGL(window_stats_finalize_kernel, dim3(B), dim3(64), 0, s, workspace, slots, (long)n, mode, coef);
  } else if (mode == PYTC_NORM_NONE && !binarize && !clip) {
    return PYTC_OK;
  }
  const long total = (long)B * n;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(window_apply_kernel, dim3(blocks), dim3(256), 0, s, x, (long)n, B, binarize, threshold, clip, coef, mode, divide);
  PYTC_LAUNCH_CHECK("window_normalize");
  return PYTC_OK;
}
