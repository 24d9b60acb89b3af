// Backward kernels of the dense-convolution (RSUNet) training step.  Correctness-first: VALU arithmetic, two-stage
// deterministic reductions (per-workgroup partial slots -> reduce in slot order), no float atomics.
//   conv3d weight gradient (one TN "GEMM" per tap over the voxel rows, shifted operand, zero padding),
//   activation derivative through the norm affine, general norm backward apply (any statistics group),
//   max-pool backward (first maximum in scan order, like torch), anisotropic depthwise conv (backward of the fixed
//   bilinear transposed-conv upsampling).
// Data gradients of the dense convs reuse the forward implicit-GEMM kernel with flipped / transposed weights.
#include "pytc_common.h"

namespace pytc {

__device__ __forceinline__ float act_fwd(float t, int act, float prm) {
  if (act == PYTC_ACT_RELU) return fmaxf(t, 0.f);
  if (act == PYTC_ACT_LEAKY) return t > 0.f ? t : prm * t;
  if (act == PYTC_ACT_ELU) return t > 0.f ? t : prm * (__expf(t) - 1.f);
  return t;
}
__device__ __forceinline__ float act_der(float t, int act, float prm) {
  if (act == PYTC_ACT_RELU) return t > 0.f ? 1.f : 0.f;
  if (act == PYTC_ACT_LEAKY) return t > 0.f ? 1.f : prm;
  if (act == PYTC_ACT_ELU) return t > 0.f ? 1.f : prm * __expf(t);
  return 1.f;
}

// ---- conv3d weight gradient ------------------------------------------------------------------------------------------
// dWp[slot][tap][o][k] = sum_{rows of slot} dY[r][o] * A[r + shift(tap)][k], A = conv input (already activated), zero
// outside the volume.  workgroup = (row slot, 64x64 (o,k) tile, tap); rows staged 32 at a time; thread = 4x4 block.
struct CwGeom { int D, H, W, kd, kh, kw; };
constexpr int CW_TO = 64, CW_TK = 64, CW_TR = 32;
template <typename T>
__global__ void __launch_bounds__(256)
conv3d_wgrad_kernel(const T* __restrict__ a, const T* __restrict__ dy, float* __restrict__ dWp, long rows_total,
                    CwGeom g, int C_in, int C_out, long rows_per_slot, int slots) {
  __shared__ float sa[CW_TR][CW_TK + 1];
  __shared__ float sd[CW_TR][CW_TO + 1];
  const int slot = blockIdx.x, tap = blockIdx.z;
  const int tiles_k = (C_in + CW_TK - 1) / CW_TK;
  const int o_base = (blockIdx.y / tiles_k) * CW_TO, k_base = (blockIdx.y % tiles_k) * CW_TK;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int tz_ = tap / (g.kh * g.kw), ty_ = (tap / g.kw) % g.kh, tx_ = tap % g.kw;
  const int dz = tz_ - g.kd / 2, dyy = ty_ - g.kh / 2, dx = tx_ - g.kw / 2;
  const long vol = (long)g.D * g.H * g.W;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const long r_begin = (long)slot * rows_per_slot;
  const long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  for (long r0 = r_begin; r0 < r_end; r0 += CW_TR) {
    for (int i = threadIdx.x; i < CW_TR * CW_TK; i += 256) {
      const int rr = i / CW_TK, kk = i % CW_TK;
      const long r = r0 + rr;
      float v = 0.f;
      if (r < r_end && k_base + kk < C_in) {
        const long rem = r % vol;
        const int x = (int)(rem % g.W), y = (int)((rem / g.W) % g.H), z = (int)(rem / ((long)g.W * g.H));
        const int sz = z + dz, sy = y + dyy, sx = x + dx;
        if (sz >= 0 && sz < g.D && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W)
          v = to_f32<T>(a[(r + ((long)dz * g.H + dyy) * g.W + dx) * C_in + k_base + kk]);
      }
      sa[rr][kk] = v;
    }
    for (int i = threadIdx.x; i < CW_TR * CW_TO; i += 256) {
      const int rr = i / CW_TO, oo = i % CW_TO;
      const long r = r0 + rr;
      sd[rr][oo] = (r < r_end && o_base + oo < C_out) ? to_f32<T>(dy[r * C_out + o_base + oo]) : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < CW_TR; ++rr) {
      float dv[4], xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { dv[i] = sd[rr][ty * 4 + i]; xv[i] = sa[rr][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], xv[j], acc[i][j]);
    }
    __syncthreads();
  }
  const int taps = g.kd * g.kh * g.kw;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o_base + ty * 4 + i;
    if (o >= C_out) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k_base + tx * 4 + j;
      if (k < C_in) dWp[(((long)slot * taps + tap) * C_out + o) * C_in + k] = acc[i][j];
    }
  }
}

__global__ void __launch_bounds__(256)
reduce_slots2_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int slots) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = 0; s < slots; ++s) a += part[(long)s * n + i];
  out[i] = a;
}

// ---- activation derivative through the norm affine: dt = da * act'(a*x + b);  dp = da * min(t, 0) (PReLU weight) ----
template <typename T>
__global__ void __launch_bounds__(256)
act_bwd_kernel(const T* __restrict__ da, const T* __restrict__ x, const float* __restrict__ ab, T* __restrict__ dt,
               T* __restrict__ dp, long rows, int C, long total, int act, float prm) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long n = i / (rows * C);
    float t = to_f32<T>(x[i]);
    if (ab) t = to_f32<T>(from_f32<T>(fmaf(t, ab[(n * 2 + 0) * C + c], ab[(n * 2 + 1) * C + c]))) ;
    const float d = to_f32<T>(da[i]);
    dt[i] = from_f32<T>(d * act_der(t, act, prm));
    if (dp) dp[i] = from_f32<T>(t < 0.f ? d * t : 0.f);
  }
}

// ---- general norm backward apply: dx = rstd * (gamma * d - M1 - xhat * M2), M per (n, c) (group-expanded means) -----
template <typename T>
__global__ void __launch_bounds__(256)
norm_bwd_apply_general_kernel(const T* __restrict__ d, const T* __restrict__ x, const float* __restrict__ mr,
                              const float* __restrict__ gamma, const float* __restrict__ M, T* __restrict__ dx,
                              long rows, int C, long total) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long n = i / (rows * C);
    const float mean = mr[(n * 2 + 0) * C + c], rstd = mr[(n * 2 + 1) * C + c];
    const float xh = (to_f32<T>(x[i]) - mean) * rstd;
    const float gd = (gamma ? gamma[c] : 1.f) * to_f32<T>(d[i]);
    dx[i] = from_f32<T>(rstd * (gd - M[(n * 2 + 0) * C + c] - xh * M[(n * 2 + 1) * C + c]));
  }
}

// ---- max-pool backward (kernel == stride == f, floor): dy goes to the first maximum of its window ------------------
template <typename T>
__global__ void __launch_bounds__(256)
maxpool3d_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int D, int H, int W, int C,
                     int fz, int fy, int fx, int Do, int Ho, int Wo, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long t = i / C;
  const int ox = (int)(t % Wo); t /= Wo;
  const int oy = (int)(t % Ho); t /= Ho;
  const int oz = (int)(t % Do);
  const long n = t / Do;
  const T* xn = x + n * (long)D * H * W * C;
  T* dn = dx + n * (long)D * H * W * C;
  float best = -INFINITY;
  long arg = -1;
  for (int a = 0; a < fz; ++a)
    for (int b = 0; b < fy; ++b)
      for (int e = 0; e < fx; ++e) {
        const long p = (((long)(oz * fz + a) * H + (oy * fy + b)) * W + (ox * fx + e)) * C + c;
        const float v = to_f32<T>(xn[p]);
        if (v > best || arg < 0) { best = v; arg = p; }       // first maximum wins (NaN-free inputs)
      }
  dn[arg] = dy[i];
}

// ---- anisotropic depthwise conv (gather): y[o][c] = sum_k x[o*s - p + k][c] * w[k][c] ------------------------------
struct DwGen { int D, H, W, C, kd, kh, kw, sz, sy, sx, pz, py, px, Do, Ho, Wo; };
template <typename T>
__global__ void __launch_bounds__(256)
dwconv3d_generic_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w, DwGen g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % g.C);
  long t = i / g.C;
  const int ox = (int)(t % g.Wo); t /= g.Wo;
  const int oy = (int)(t % g.Ho); t /= g.Ho;
  const int oz = (int)(t % g.Do);
  const long n = t / g.Do;
  const T* xn = x + n * (long)g.D * g.H * g.W * g.C;
  float acc = 0.f;
  for (int a = 0; a < g.kd; ++a) {
    const int iz = oz * g.sz - g.pz + a;
    if (iz < 0 || iz >= g.D) continue;
    for (int b = 0; b < g.kh; ++b) {
      const int iy = oy * g.sy - g.py + b;
      if (iy < 0 || iy >= g.H) continue;
      for (int e = 0; e < g.kw; ++e) {
        const int ix = ox * g.sx - g.px + e;
        if (ix < 0 || ix >= g.W) continue;
        acc = fmaf(to_f32<T>(xn[(((long)iz * g.H + iy) * g.W + ix) * g.C + c]), w[((long)(a * g.kh + b) * g.kw + e) * g.C + c], acc);
      }
    }
  }
  y[i] = from_f32<T>(acc);
}

static int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 16384 ? b : 16384); }

}  // namespace pytc

using namespace pytc;

#define RS_DISPATCH(dtype, BF, F32, what)                                   \
  if (dtype == PYTC_BF16) { BF; } else if (dtype == PYTC_F32) { F32; } else { \
    set_error(what ": bad dtype %d", dtype); return PYTC_ERR_INVALID; }

extern "C" int pytc_conv3d_wgrad_slots(int64_t rows_total) {
  const long s = rows_total / 8192;
  return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

extern "C" int pytc_conv3d_wgrad(const void* a, const void* dy, float* dW, float* workspace, int N, int D, int H, int W,
                                 int C_in, int C_out, const int32_t* kernel, int dtype, void* stream) {
  PYTC_REQUIRE(a && dy && dW && workspace && kernel && N >= 1, "conv3d_wgrad: bad arguments");
  CwGeom g{D, H, W, kernel[0], kernel[1], kernel[2]};
  PYTC_REQUIRE(g.kd % 2 == 1 && g.kh % 2 == 1 && g.kw % 2 == 1, "conv3d_wgrad: odd kernel sizes only ('same' padding)");
  const long rows_total = (long)N * D * H * W;
  const int slots = pytc_conv3d_wgrad_slots(rows_total);
  const long rps = (rows_total + slots - 1) / slots;
  const int taps = g.kd * g.kh * g.kw;
  dim3 grid(slots, ((C_out + CW_TO - 1) / CW_TO) * ((C_in + CW_TK - 1) / CW_TK), taps), block(256);
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(conv3d_wgrad_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)a, (const bf16_t*)dy, workspace, rows_total, g, C_in, C_out, rps, slots),
              hipLaunchKernelGGL(conv3d_wgrad_kernel<float>, grid, block, 0, s, (const float*)a, (const float*)dy, workspace, rows_total, g, C_in, C_out, rps, slots),
              "conv3d_wgrad")
  const long nW = (long)taps * C_out * C_in;
  hipLaunchKernelGGL(reduce_slots2_kernel, dim3(ceil_div(nW, 256)), dim3(256), 0, s, workspace, dW, nW, slots);
  PYTC_LAUNCH_CHECK("conv3d_wgrad");
  return PYTC_OK;
}

extern "C" int pytc_act_bwd(const void* da, const void* x, const float* ab, void* dt, void* dp, int N, int64_t rows, int C,
                            int act, float prm, int dtype, void* stream) {
  PYTC_REQUIRE(da && x && dt && N >= 1 && rows >= 1 && C >= 1, "act_bwd: bad arguments");
  const long total = (long)N * rows * C;
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)da, (const bf16_t*)x, ab, (bf16_t*)dt, (bf16_t*)dp, (long)rows, C, total, act, prm),
              hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)da, (const float*)x, ab, (float*)dt, (float*)dp, (long)rows, C, total, act, prm),
              "act_bwd")
  PYTC_LAUNCH_CHECK("act_bwd");
  return PYTC_OK;
}

extern "C" int pytc_norm_bwd_apply_general(const void* d, const void* x, const float* mean_rstd, const float* gamma,
                                           const float* M, void* dx, int N, int64_t rows, int C, int dtype, void* stream) {
  PYTC_REQUIRE(d && x && mean_rstd && M && dx, "norm_bwd_apply_general: null pointer");
  const long total = (long)N * rows * C;
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(norm_bwd_apply_general_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)d, (const bf16_t*)x, mean_rstd, gamma, M, (bf16_t*)dx, (long)rows, C, total),
              hipLaunchKernelGGL(norm_bwd_apply_general_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)d, (const float*)x, mean_rstd, gamma, M, (float*)dx, (long)rows, C, total),
              "norm_bwd_apply_general")
  PYTC_LAUNCH_CHECK("norm_bwd_apply_general");
  return PYTC_OK;
}

extern "C" int pytc_maxpool3d_bwd(const void* x, const void* dy, void* dx, int N, int D, int H, int W, int C, int fz, int fy,
                                  int fx, int dtype, void* stream) {
  PYTC_REQUIRE(x && dy && dx && fz >= 1 && fy >= 1 && fx >= 1, "maxpool3d_bwd: bad arguments");
  const int Do = D / fz, Ho = H / fy, Wo = W / fx;
  const long total = (long)N * Do * Ho * Wo * C;
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)N * D * H * W * C * (dtype == PYTC_BF16 ? 2 : 4);
  if (hipMemsetAsync(dx, 0, bytes, s) != hipSuccess) { set_error("maxpool3d_bwd: memset failed"); return PYTC_ERR_HIP; }
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(maxpool3d_bwd_kernel<bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, D, H, W, C, fz, fy, fx, Do, Ho, Wo, total),
              hipLaunchKernelGGL(maxpool3d_bwd_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)dx, D, H, W, C, fz, fy, fx, Do, Ho, Wo, total),
              "maxpool3d_bwd")
  PYTC_LAUNCH_CHECK("maxpool3d_bwd");
  return PYTC_OK;
}

extern "C" int pytc_dwconv3d_generic_fwd(const void* x, void* y, const float* w, int N, int D, int H, int W, int C,
                                         const int32_t* kernel, const int32_t* stride, const int32_t* pad,
                                         const int32_t* out_dims, int dtype, void* stream) {
  PYTC_REQUIRE(x && y && w && kernel && stride && pad && out_dims, "dwconv3d_generic: null pointer");
  DwGen g{D, H, W, C, kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
          out_dims[0], out_dims[1], out_dims[2]};
  const long total = (long)N * g.Do * g.Ho * g.Wo * C;
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(dwconv3d_generic_kernel<bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, w, g, total),
              hipLaunchKernelGGL(dwconv3d_generic_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)x, (float*)y, w, g, total),
              "dwconv3d_generic")
  PYTC_LAUNCH_CHECK("dwconv3d_generic");
  return PYTC_OK;
}
