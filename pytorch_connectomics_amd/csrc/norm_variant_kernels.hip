// Backward kernels of the MedNeXt block variants (reference constructor mednext_models.py:449-463: norm_type='layer',
// grn=True; upstream nnunet_mednext blocks.py LayerNorm(channels_first) and the GRN branch of MedNeXtBlock.forward):
//
//   layernorm_rows_bwd   per-voxel LayerNorm over C: dx, and slot partials of dgamma / dbeta
//   grn_bwd_apply        dhp = (dh2 * A[n][c] + gelu(hp) * B[n][c]) * gelu'(hp): the GRN and GELU derivatives in one pass
//
// Both are HBM-bound elementwise / row passes (3 tensor reads-or-writes per element); the (n, c) GRN coefficients are built
// on the host side from the (N, 2, C) sums pytc_norm_bwd_stats already provides.
#include "pw_common.h"

namespace pytc {

// d/dx [x * Phi(x)] with libm erff -- the same derivative gelu_kernel / RES_GELU_BWD apply (train_kernels.hip gelu_grad)
__device__ __forceinline__ float gelu_grad_erf(float x) {
  const float phi_big = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return phi_big + x * pdf;
}

// L = C / VEC lanes share a row (power of two <= 64, same mapping as layernorm_rows_kernel).  Per row:
//   xhat = (x - mean) * rstd,  g = dy * gamma,  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))
// Every lane also carries the running sums of dy (-> dbeta) and dy * xhat (-> dgamma) of its VEC channels over the rows
// it visits; the row groups of a workgroup meet in LDS and each workgroup writes ONE slot of partial[slot][2][C]
// ([0] = sum dy, [1] = sum dy * xhat), reduced across slots by pytc_reduce_slots(_multi): deterministic.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
layernorm_rows_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                          T* __restrict__ dx, float* __restrict__ partial, long rows, int C, float eps) {
  extern __shared__ float lds[];                           // [rows_per_block][2][C]
  const int L = C / VEC;
  const int rows_per_block = 256 / L;
  const int lr = threadIdx.x / L, lc = threadIdx.x % L;
  float gm[VEC], acc_b[VEC], acc_g[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    gm[j] = gamma ? gamma[lc * VEC + j] : 1.0f;
    acc_b[j] = 0.f;
    acc_g[j] = 0.f;
  }
  const float inv_c = 1.0f / (float)C;
  for (long r = (long)blockIdx.x * rows_per_block + lr; r < rows; r += (long)gridDim.x * rows_per_block) {
    float v[VEC], d[VEC];
    VecIO<T, VEC>::load(x + r * C + lc * VEC, v);
    VecIO<T, VEC>::load(dy + r * C + lc * VEC, d);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += v[j];
    for (int off = 1; off < L; off <<= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const float c = v[j] - mean; q = fmaf(c, c, q); }
    for (int off = 1; off < L; off <<= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      v[j] = (v[j] - mean) * rstd;                         // xhat
      acc_b[j] += d[j];
      acc_g[j] = fmaf(d[j], v[j], acc_g[j]);
      d[j] *= gm[j];                                       // g
      s1 += d[j];
      s2 = fmaf(d[j], v[j], s2);
    }
    for (int off = 1; off < L; off <<= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    const float m1 = s1 * inv_c, m2 = s2 * inv_c;
#pragma unroll
    for (int j = 0; j < VEC; ++j) d[j] = rstd * (d[j] - m1 - v[j] * m2);
    VecIO<T, VEC>::store(dx + r * C + lc * VEC, d);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    lds[(lr * 2 + 0) * C + lc * VEC + j] = acc_b[j];
    lds[(lr * 2 + 1) * C + lc * VEC + j] = acc_g[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float a = 0.f;
    for (int g = 0; g < rows_per_block; ++g) a += lds[g * 2 * C + i];
    partial[(long)blockIdx.x * 2 * C + i] = a;
  }
}

// dhp[n][r][c] = (dh2 * A[n][c] + gelu(hp) * B[n][c]) * gelu'(hp)
template <typename T>
__global__ void __launch_bounds__(256)
grn_bwd_apply_kernel(const T* __restrict__ dh2, const T* __restrict__ hp, const float* __restrict__ A,
                     const float* __restrict__ B, T* __restrict__ out, long rows, int C, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long n = i / (rows * C);
    const float p = to_f32<T>(hp[i]);
    // the forward's h is gelu(hp) ROUNDED to the storage type (pytc_gelu writes T): use the same value here
    const float h = to_f32<T>(from_f32<T>(gelu_erf(p)));
    const float dh = fmaf(to_f32<T>(dh2[i]), A[n * C + c], h * B[n * C + c]);
    out[i] = from_f32<T>(dh * gelu_grad_erf(p));
  }
}

static int ln_vec(int C, int dtype) {
  auto pow2 = [](int v) { return v >= 1 && (v & (v - 1)) == 0; };
  for (int v = 8; v >= 1; v >>= 1)
    if (C % v == 0 && pow2(C / v) && C / v <= 64) return v;
  (void)dtype;
  return 0;
}

static long ln_bwd_blocks(long rows, int C, int vec) {
  const int rpb = 256 / (C / vec);
  long blocks = (rows + rpb - 1) / rpb;
  return blocks > 1024 ? 1024 : blocks;                     // = slots of the partial buffer
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_layernorm_rows_bwd_slots(int64_t rows, int C, int dtype) {
  const int vec = ln_vec(C, dtype);
  if (!vec || rows < 1) return 0;
  return (int)ln_bwd_blocks((long)rows, C, vec);
}

extern "C" int pytc_layernorm_rows_bwd(const void* dy, const void* x, const float* gamma, void* dx, float* partial,
                                       int64_t rows, int C, float eps, int dtype, void* stream) {
  PYTC_REQUIRE(dy && x && dx && partial && rows >= 1 && C >= 1, "layernorm_rows_bwd: bad arguments");
  const int vec = ln_vec(C, dtype);
  PYTC_REQUIRE(vec > 0, "layernorm_rows_bwd: C=%d must be VEC * 2^k with VEC <= 8 and 2^k <= 64", C);
  hipStream_t s = (hipStream_t)stream;
  const long blocks = ln_bwd_blocks((long)rows, C, vec);
  const size_t lds = (size_t)(256 / (C / vec)) * 2 * C * sizeof(float);       // = 256 * VEC * 8 bytes <= 16 KB
#define LNB_LAUNCH(TT, V)                                                                                          \
  hipLaunchKernelGGL((layernorm_rows_bwd_kernel<TT, V>), dim3((unsigned)blocks), dim3(256), lds, s, (const TT*)dy, \
                     (const TT*)x, gamma, (TT*)dx, partial, (long)rows, C, eps)
#define LNB_VEC(TT)                         \
  switch (vec) {                            \
    case 8: LNB_LAUNCH(TT, 8); break;       \
    case 4: LNB_LAUNCH(TT, 4); break;       \
    case 2: LNB_LAUNCH(TT, 2); break;       \
    default: LNB_LAUNCH(TT, 1); break;      \
  }
  if (dtype == PYTC_BF16) { LNB_VEC(bf16_t) }
  else if (dtype == PYTC_F32) { LNB_VEC(float) }
  else { PYTC_REQUIRE(false, "layernorm_rows_bwd: bad dtype"); }
#undef LNB_VEC
#undef LNB_LAUNCH
  PYTC_LAUNCH_CHECK("layernorm_rows_bwd");
  return PYTC_OK;
}

extern "C" int pytc_grn_bwd_apply(const void* dh2, const void* hp, const float* A, const float* B, void* out, int N,
                                  int64_t rows, int C, int dtype, void* stream) {
  PYTC_REQUIRE(dh2 && hp && A && B && out && N >= 1 && rows >= 1 && C >= 1, "grn_bwd_apply: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * rows * C;
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(grn_bwd_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)dh2,
                       (const bf16_t*)hp, A, B, (bf16_t*)out, (long)rows, C, total);
  else if (dtype == PYTC_F32)
    hipLaunchKernelGGL(grn_bwd_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dh2,
                       (const float*)hp, A, B, (float*)out, (long)rows, C, total);
  else
    PYTC_REQUIRE(false, "grn_bwd_apply: bad dtype");
  PYTC_LAUNCH_CHECK("grn_bwd_apply");
  return PYTC_OK;
}
