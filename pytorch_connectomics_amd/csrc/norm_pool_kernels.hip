// Per-(n,c) statistics of an arbitrary NDHWC tensor, max-pooling and the generic depthwise transposed conv
// (RSUNet's fixed-weight "bilinear" upsampling) -- HBM-bound elementwise / reduction kernels, lanes along C.
#include "pytc_common.h"
#include "colstats.h"

namespace pytc {

// ---- channel statistics: stats[N][slots][2][C] partial (sum, sumsq), reduced later in fixed order --------------
template <typename T>
__global__ void __launch_bounds__(256)
channel_stats_kernel(const T* __restrict__ x, float* __restrict__ stats, long rows, int C, int slots, long rows_per_slot) {
  extern __shared__ float lds[];   // [rows_in_flight][2][Cc] with Cc = min(C, 256)
  const int n = blockIdx.y, slot = blockIdx.x;
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
  const T* xn = x + (long)n * rows * C;
  // thread -> (channel c = tid % Cw, row lane rl = tid / Cw) for each channel window of width Cw <= 256
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int Cw = (C - c0) < 256 ? (C - c0) : 256;
    const int RL = 256 / Cw;
    const int c = threadIdx.x % Cw, rl = threadIdx.x / Cw;
    float s1 = 0.f, s2 = 0.f;
    if (rl < RL) {
      for (long r = r0 + rl; r < r1; r += RL) {
        float v = to_f32<T>(xn[r * C + c0 + c]);
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
      lds[(rl * 2 + 0) * Cw + c] = s1;
      lds[(rl * 2 + 1) * Cw + c] = s2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cw; i += blockDim.x) {
      const int which = i / Cw, ch = i % Cw;
      float a = 0.f;
      for (int q = 0; q < RL; ++q) a += lds[(q * 2 + which) * Cw + ch];
      stats[(((long)n * slots + slot) * 2 + which) * C + c0 + ch] = a;
    }
    __syncthreads();
  }
}

// ---- norm finalize with channel groups: ab[N][2][C] ------------------------------------------------------------
__global__ void __launch_bounds__(256)
norm_finalize_groups_kernel(const float* __restrict__ stats, int slots, float count, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float eps, int groups, float* __restrict__ ab, int C,
                            float* __restrict__ mr = nullptr, int cpg_arg = 0) {
  // one workgroup per (n, group); deterministic two-level reduction
  // cpg_arg > 0: `groups` groups of cpg_arg channels each; the channels from groups * cpg_arg on are alignment padding of an
  // all-zero tensor tail (the extra workgroup blockIdx.x == groups): affine (0, 0), mean / rstd 0 -- they stay zero through
  // any activation with act(0) = 0 and contribute nothing to the backward pass
  __shared__ float red[2][256];
  const int n = blockIdx.y, g = blockIdx.x;
  const int cpg = cpg_arg > 0 ? cpg_arg : C / groups;
  if (g == groups) {
    for (int c = groups * cpg + threadIdx.x; c < C; c += blockDim.x) {
      ab[((long)n * 2 + 0) * C + c] = 0.f;
      ab[((long)n * 2 + 1) * C + c] = 0.f;
      if (mr) { mr[((long)n * 2 + 0) * C + c] = 0.f; mr[((long)n * 2 + 1) * C + c] = 0.f; }
    }
    return;
  }
  float a1 = 0.f, a2 = 0.f;
  const float* base = stats + (long)n * slots * 2 * C;
  // same order of additions as the plain loop, eight slot rows in flight per thread (1 024 slots x 6 channels were 24 dependent round
  // trips: 15 us for a launch that moves 50 KB; RSUNet runs 37 of them per training step)
  const long total = (long)slots * cpg;
  long i = threadIdx.x;
  for (; i + 7L * blockDim.x < total; i += 8L * blockDim.x) {
    float v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long j = i + (long)u * blockDim.x;
      const int s = (int)(j / cpg), c = g * cpg + (int)(j % cpg);
      v1[u] = base[((long)s * 2 + 0) * C + c];
      v2[u] = base[((long)s * 2 + 1) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a1 += v1[u]; a2 += v2[u]; }
  }
  for (; i < total; i += blockDim.x) {
    const int s = (int)(i / cpg), c = g * cpg + (int)(i % cpg);
    a1 += base[((long)s * 2 + 0) * C + c];
    a2 += base[((long)s * 2 + 1) * C + c];
  }
  red[0][threadIdx.x] = a1;
  red[1][threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t1 = 0.f, t2 = 0.f;
    for (int i = 0; i < 256; ++i) { t1 += red[0][i]; t2 += red[1][i]; }
    const float cnt = count * cpg;
    const float mean = t1 / cnt;
    const float var = fmaxf(t2 / cnt - mean * mean, 0.f);
    float rstd = rsqrtf(var + eps);
    rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
    red[0][0] = mean;
    red[1][0] = rstd;
  }
  __syncthreads();
  const float mean = red[0][0], rstd = red[1][0];
  for (int i = threadIdx.x; i < cpg; i += blockDim.x) {
    const int c = g * cpg + i;
    const float a = (gamma ? gamma[c] : 1.f) * rstd;
    ab[((long)n * 2 + 0) * C + c] = a;
    ab[((long)n * 2 + 1) * C + c] = (beta ? beta[c] : 0.f) - mean * a;
    if (mr) {
      mr[((long)n * 2 + 0) * C + c] = mean;
      mr[((long)n * 2 + 1) * C + c] = rstd;
    }
  }
}

// ---- max pooling, kernel == stride == (fz,fy,fx), floor semantics of nn.MaxPool3d -------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
maxpool3d_kernel(const T* __restrict__ x, T* __restrict__ y, int D, int H, int W, int C, int fz, int fy, int fx,
                 int Do, int Ho, int Wo, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long t = i / C;
  const int ox = (int)(t % Wo); t /= Wo;
  const int oy = (int)(t % Ho); t /= Ho;
  const int oz = (int)(t % Do);
  const long n = t / Do;
  const T* xn = x + n * (long)D * H * W * C;
  float m = -3.402823466e38f;
  for (int a = 0; a < fz; ++a)
    for (int b = 0; b < fy; ++b)
      for (int d = 0; d < fx; ++d)
        m = fmaxf(m, to_f32<T>(xn[(((long)(oz * fz + a) * H + (oy * fy + b)) * W + (ox * fx + d)) * C + c]));
  y[i] = from_f32<T>(m);
}

// ---- generic depthwise transposed conv (gather form): out = (in-1)*s - 2p + k per axis --------------------------
struct DwTGen {
  int D, H, W, C, kd, kh, kw, sz, sy, sx, pz, py, px, Do, Ho, Wo;
};

template <typename T>
__global__ void __launch_bounds__(256)
dwconvT3d_generic_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w, DwTGen g, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % g.C);
  long t = i / g.C;
  const int ox = (int)(t % g.Wo); t /= g.Wo;
  const int oy = (int)(t % g.Ho); t /= g.Ho;
  const int oz = (int)(t % g.Do);
  const long n = t / g.Do;
  const T* xn = x + n * (long)g.D * g.H * g.W * g.C;
  float acc = 0.f;
  for (int kz = 0; kz < g.kd; ++kz) {
    const int tz = oz + g.pz - kz;
    if (tz < 0 || tz % g.sz || tz / g.sz >= g.D) continue;
    for (int ky = 0; ky < g.kh; ++ky) {
      const int ty = oy + g.py - ky;
      if (ty < 0 || ty % g.sy || ty / g.sy >= g.H) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int tx = ox + g.px - kx;
        if (tx < 0 || tx % g.sx || tx / g.sx >= g.W) continue;
        acc = fmaf(to_f32<T>(xn[(((long)(tz / g.sz) * g.H + ty / g.sy) * g.W + tx / g.sx) * g.C + c]),
                   w[((long)(kz * g.kh + ky) * g.kw + kx) * g.C + c], acc);
      }
    }
  }
  y[i] = from_f32<T>(acc);
}

// vector form: one lane = VEC channels (16 bytes) of one output voxel, looping only over the taps that land on an input
// voxel (k = (o + p) mod s, + s, ...: ascending like the scalar loop, so the sums are bit-identical to it)
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
dwconvT3d_generic_vec_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w, DwTGen g, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int CV = g.C / VEC;
  const int c = (int)(i % CV) * VEC;
  long t = i / CV;
  const int ox = (int)(t % g.Wo); t /= g.Wo;
  const int oy = (int)(t % g.Ho); t /= g.Ho;
  const int oz = (int)(t % g.Do);
  const long n = t / g.Do;
  const T* xn = x + n * (long)g.D * g.H * g.W * g.C;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  for (int kz = (oz + g.pz) % g.sz; kz < g.kd; kz += g.sz) {
    const int iz = (oz + g.pz - kz) / g.sz;
    if (oz + g.pz - kz < 0 || iz >= g.D) continue;
    for (int ky = (oy + g.py) % g.sy; ky < g.kh; ky += g.sy) {
      const int iy = (oy + g.py - ky) / g.sy;
      if (oy + g.py - ky < 0 || iy >= g.H) continue;
      for (int kx = (ox + g.px) % g.sx; kx < g.kw; kx += g.sx) {
        const int ix = (ox + g.px - kx) / g.sx;
        if (ox + g.px - kx < 0 || ix >= g.W) continue;
        float xv[VEC];
        VecIO<T, VEC>::load(xn + (((long)iz * g.H + iy) * g.W + ix) * g.C + c, xv);
        const float* wp = w + ((long)(kz * g.kh + ky) * g.kw + kx) * g.C + c;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(xv[j], wp[j], acc[j]);
      }
    }
  }
  VecIO<T, VEC>::store(y + (((n * g.Do + oz) * g.Ho + oy) * g.Wo + ox) * g.C + c, acc);
}

// ---- channels-first LayerNorm of MedNeXt (norm_type='layer'): every voxel row normalised over its C channels ---------
// y[r][c] = gamma[c] * (x[r][c] - mean_r) / sqrt(var_r + eps) + beta[c]   (biased variance, two-pass in registers).
// L = C / VEC lanes share a row (power of two <= 64), their partial sums meet through xor-shuffles.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
layernorm_rows_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                      const float* __restrict__ beta, long rows, int C, float eps) {
  const int L = C / VEC;                                  // lanes per row
  const int rows_per_block = 256 / L;
  const int lr = threadIdx.x / L, lc = threadIdx.x % L;
  for (long r = (long)blockIdx.x * rows_per_block + lr; r < rows; r += (long)gridDim.x * rows_per_block) {
    float v[VEC];
    VecIO<T, VEC>::load(x + r * C + lc * VEC, v);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += v[j];
    for (int off = 1; off < L; off <<= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const float d = v[j] - mean; q = fmaf(d, d, q); }
    for (int off = 1; off < L; off <<= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = lc * VEC + j;
      v[j] = fmaf((v[j] - mean) * rstd, gamma ? gamma[c] : 1.0f, beta ? beta[c] : 0.f);
    }
    VecIO<T, VEC>::store(y + r * C + lc * VEC, v);
  }
}

// ---- y = act(a[n][c]*x + b[n][c]) elementwise (norm-apply + activation when it cannot ride in a conv prologue) ----
template <typename T>
__global__ void __launch_bounds__(256)
affine_act_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ ab, long rows, int C, int act,
                  float prm, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long n = i / ((long)rows * C);
    float v = to_f32<T>(x[i]);
    if (ab) v = fmaf(v, ab[(n * 2 + 0) * C + c], ab[(n * 2 + 1) * C + c]);
    if (act == PYTC_ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == PYTC_ACT_LEAKY) v = v > 0.f ? v : v * prm;
    else if (act == PYTC_ACT_ELU) v = v > 0.f ? v : prm * (__expf(v) - 1.0f);
    else if (act == PYTC_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
    else if (act == PYTC_ACT_TANH) v = tanhf(v);
    y[i] = from_f32<T>(v);
  }
}

// 16-byte form (C % VEC == 0): one thread = VEC consecutive channels of one voxel, one index decomposition per chunk instead of
// a 64-bit modulo and divide per element; same arithmetic per element as affine_act_kernel.
template <typename T>
__global__ void __launch_bounds__(256)
affine_act_vec_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ ab, long rows, int C, int act,
                      float prm, long chunks) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cq = C / VEC;
  const long per_sample = rows * cq;
  long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; q < chunks; q += stride) {
    const int c0 = (int)(q % cq) * VEC;
    const long n = q / per_sample;
    float v[VEC];
    VecIO<T, VEC>::load(x + q * VEC, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = v[j];
      if (ab) t = fmaf(t, ab[(n * 2 + 0) * C + c0 + j], ab[(n * 2 + 1) * C + c0 + j]);
      if (act == PYTC_ACT_RELU) t = fmaxf(t, 0.f);
      else if (act == PYTC_ACT_LEAKY) t = t > 0.f ? t : t * prm;
      else if (act == PYTC_ACT_ELU) t = t > 0.f ? t : prm * (__expf(t) - 1.0f);
      else if (act == PYTC_ACT_SIGMOID) t = 1.f / (1.f + __expf(-t));
      else if (act == PYTC_ACT_TANH) t = tanhf(t);
      v[j] = t;
    }
    VecIO<T, VEC>::store(y + q * VEC, v);
  }
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_affine_act(const void* x, void* y, const float* ab, int N, int64_t rows, int C, int act, float prm,
                               int dtype, void* stream) {
  PYTC_REQUIRE(x && y && N >= 1 && rows >= 1 && C >= 1, "affine_act: bad arguments");
  const long total = (long)N * rows * C;
  const int vec = dtype == PYTC_BF16 ? 8 : 4;
  if ((dtype == PYTC_BF16 || dtype == PYTC_F32) && C % vec == 0 && tuning_get("elementwise_vec", 1)) {
    const long chunks = total / vec;
    const int vb = (int)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
    if (dtype == PYTC_BF16)
      hipLaunchKernelGGL(affine_act_vec_kernel<bf16_t>, dim3(vb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, ab,
                         (long)rows, C, act, prm, chunks);
    else
      hipLaunchKernelGGL(affine_act_vec_kernel<float>, dim3(vb), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, ab,
                         (long)rows, C, act, prm, chunks);
    PYTC_LAUNCH_CHECK("affine_act");
    return PYTC_OK;
  }
  int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(affine_act_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, ab, (long)rows, C, act, prm, total);
  else if (dtype == PYTC_F32)
    hipLaunchKernelGGL(affine_act_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (float*)y, ab, (long)rows, C, act, prm, total);
  else
    PYTC_REQUIRE(false, "affine_act: bad dtype");
  PYTC_LAUNCH_CHECK("affine_act");
  return PYTC_OK;
}

extern "C" int pytc_channel_stats_slots(int64_t rows) { return colstats_slots(rows); }

extern "C" int pytc_channel_stats(const void* x, float* stats, int N, int64_t rows, int C, int dtype, void* stream) {
  PYTC_REQUIRE(x && stats && N >= 1 && rows >= 1 && C >= 1, "channel_stats: bad arguments");
  const int slots = pytc_channel_stats_slots(rows);
  const long rps = (rows + slots - 1) / slots;
  const int Cw = C < 256 ? C : 256;
  size_t lds = (size_t)(256 / Cw) * 2 * Cw * sizeof(float);
  dim3 grid(slots, N), block(256);
  if (dtype == PYTC_BF16 && C % 8 == 0)
    hipLaunchKernelGGL((colstats_kernel<bf16_t, 0>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)nullptr,
                       (const float*)nullptr, stats, (long)rows, C, slots, rps);
  else if (dtype == PYTC_F32 && C % 4 == 0)
    hipLaunchKernelGGL((colstats_kernel<float, 0>), grid, block, 0, (hipStream_t)stream, (const float*)x, (const float*)nullptr,
                       (const float*)nullptr, stats, (long)rows, C, slots, rps);
  else if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, grid, block, lds, (hipStream_t)stream, (const bf16_t*)x, stats,
                       (long)rows, C, slots, rps);
  else if (dtype == PYTC_F32)
    hipLaunchKernelGGL(channel_stats_kernel<float>, grid, block, lds, (hipStream_t)stream, (const float*)x, stats,
                       (long)rows, C, slots, rps);
  else
    PYTC_REQUIRE(false, "channel_stats: bad dtype");
  PYTC_LAUNCH_CHECK("channel_stats");
  return PYTC_OK;
}

extern "C" int pytc_norm_finalize_groups(const float* stats, int slots, float count, const float* gamma,
                                         const float* beta, float eps, int groups, float* ab, int N, int C,
                                         void* stream) {
  PYTC_REQUIRE(stats && ab && slots >= 1 && count > 0 && groups >= 1 && C % groups == 0, "norm_finalize_groups: bad arguments");
  hipLaunchKernelGGL(norm_finalize_groups_kernel, dim3(groups, N), dim3(256), 0, (hipStream_t)stream, stats, slots,
                     count, gamma, beta, eps, groups, ab, C);
  PYTC_LAUNCH_CHECK("norm_finalize_groups");
  return PYTC_OK;
}

extern "C" int pytc_norm_finalize_groups_cpg(const float* stats, int slots, float count, const float* gamma, const float* beta,
                                             float eps, int groups, int cpg, float* ab, float* mr, int N, int C, void* stream) {
  PYTC_REQUIRE(stats && ab && slots >= 1 && count > 0 && groups >= 1, "norm_finalize_groups: bad arguments");
  PYTC_REQUIRE(cpg > 0 ? groups * cpg <= C : C % groups == 0,
               "norm_finalize_groups: C=%d, groups=%d, channels per group %d", C, groups, cpg);
  const int pad = (cpg > 0 && groups * cpg < C) ? 1 : 0;
  hipLaunchKernelGGL(norm_finalize_groups_kernel, dim3(groups + pad, N), dim3(256), 0, (hipStream_t)stream, stats, slots,
                     count, gamma, beta, eps, groups, ab, C, mr, cpg);
  PYTC_LAUNCH_CHECK("norm_finalize_groups");
  return PYTC_OK;
}

extern "C" int pytc_norm_finalize_groups_mr(const float* stats, int slots, float count, const float* gamma,
                                            const float* beta, float eps, int groups, float* ab, float* mr, int N, int C,
                                            void* stream) {
  PYTC_REQUIRE(mr, "norm_finalize_groups_mr: null mean/rstd output");
  return pytc_norm_finalize_groups_cpg(stats, slots, count, gamma, beta, eps, groups, 0, ab, mr, N, C, stream);
}

extern "C" int pytc_maxpool3d_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                                  int dtype, void* stream) {
  PYTC_REQUIRE(x && y && fz >= 1 && fy >= 1 && fx >= 1, "maxpool3d: bad arguments");
  const int Do = D / fz, Ho = H / fy, Wo = W / fx;
  PYTC_REQUIRE(Do >= 1 && Ho >= 1 && Wo >= 1, "maxpool3d: output would be empty");
  const long total = (long)N * Do * Ho * Wo * C;
  dim3 grid(ceil_div(total, 256)), block(256);
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(maxpool3d_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, D,
                       H, W, C, fz, fy, fx, Do, Ho, Wo, total);
  else if (dtype == PYTC_F32)
    hipLaunchKernelGGL(maxpool3d_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)y, D, H,
                       W, C, fz, fy, fx, Do, Ho, Wo, total);
  else
    PYTC_REQUIRE(false, "maxpool3d: bad dtype");
  PYTC_LAUNCH_CHECK("maxpool3d");
  return PYTC_OK;
}

extern "C" int pytc_dwconvT3d_generic_fwd(const void* x, void* y, const float* w, int N, int D, int H, int W, int C,
                                          const int32_t* kernel, const int32_t* stride, const int32_t* pad, int dtype,
                                          void* stream) {
  PYTC_REQUIRE(x && y && w && kernel && stride && pad, "dwconvT3d_generic: null pointer");
  DwTGen g;
  g.D = D; g.H = H; g.W = W; g.C = C;
  g.kd = kernel[0]; g.kh = kernel[1]; g.kw = kernel[2];
  g.sz = stride[0]; g.sy = stride[1]; g.sx = stride[2];
  g.pz = pad[0]; g.py = pad[1]; g.px = pad[2];
  PYTC_REQUIRE(g.sz >= 1 && g.sy >= 1 && g.sx >= 1 && g.kd >= 1 && g.kh >= 1 && g.kw >= 1, "dwconvT3d_generic: bad geometry");
  g.Do = (D - 1) * g.sz - 2 * g.pz + g.kd; g.Ho = (H - 1) * g.sy - 2 * g.py + g.kh; g.Wo = (W - 1) * g.sx - 2 * g.px + g.kw;
  PYTC_REQUIRE(g.Do >= 1 && g.Ho >= 1 && g.Wo >= 1, "dwconvT3d_generic: empty output");
  const long total = (long)N * g.Do * g.Ho * g.Wo * C;
  dim3 grid(ceil_div(total, 256)), block(256);
  if (dtype == PYTC_BF16 && C % 8 == 0)
    hipLaunchKernelGGL((dwconvT3d_generic_vec_kernel<bf16_t, 8>), dim3(ceil_div(total / 8, 256)), block, 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, w, g, total / 8);
  else if (dtype == PYTC_F32 && C % 4 == 0)
    hipLaunchKernelGGL((dwconvT3d_generic_vec_kernel<float, 4>), dim3(ceil_div(total / 4, 256)), block, 0, (hipStream_t)stream,
                       (const float*)x, (float*)y, w, g, total / 4);
  else if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(dwconvT3d_generic_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, w, g, total);
  else if (dtype == PYTC_F32)
    hipLaunchKernelGGL(dwconvT3d_generic_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)y,
                       w, g, total);
  else
    PYTC_REQUIRE(false, "dwconvT3d_generic: bad dtype");
  PYTC_LAUNCH_CHECK("dwconvT3d_generic");
  return PYTC_OK;
}

extern "C" int pytc_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int C,
                                   float eps, int dtype, void* stream) {
  PYTC_REQUIRE(x && y && rows >= 1 && C >= 1, "layernorm_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  // widest vector such that C / VEC is a power of two <= 64 (shuffle groups stay inside a wave and tile the workgroup)
  auto pow2 = [](int v) { return v >= 1 && (v & (v - 1)) == 0; };
  int vec = 0;          // fp32 rows use 8-wide chunks too (two 16-byte loads): C = 512 still fits the 64 lanes of a wave
  for (int v = 8; v >= 1; v >>= 1)
    if (C % v == 0 && pow2(C / v) && C / v <= 64) { vec = v; break; }
  PYTC_REQUIRE(vec > 0, "layernorm_rows: C=%d must be VEC * 2^k with 2^k <= 64", C);
  const int rpb = 256 / (C / vec);
  long blocks = (rows + rpb - 1) / rpb;
  if (blocks > 65536) blocks = 65536;
#define LN_LAUNCH(TT, V) hipLaunchKernelGGL((layernorm_rows_kernel<TT, V>), dim3((unsigned)blocks), dim3(256), 0, s, (const TT*)x, (TT*)y, gamma, beta, (long)rows, C, eps)
  if (dtype == PYTC_BF16) {
    if (vec == 8) LN_LAUNCH(bf16_t, 8); else if (vec == 4) LN_LAUNCH(bf16_t, 4); else if (vec == 2) LN_LAUNCH(bf16_t, 2); else LN_LAUNCH(bf16_t, 1);
  } else if (dtype == PYTC_F32) {
    if (vec == 8) LN_LAUNCH(float, 8); else if (vec == 4) LN_LAUNCH(float, 4); else if (vec == 2) LN_LAUNCH(float, 2); else LN_LAUNCH(float, 1);
  } else {
    PYTC_REQUIRE(false, "layernorm_rows: bad dtype");
  }
#undef LN_LAUNCH
  PYTC_LAUNCH_CHECK("layernorm_rows");
  return PYTC_OK;
}
