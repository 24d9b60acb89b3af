// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of the PyTC hot path.
// NDHWC activations, fp32 or bf16 storage, fp32 accumulation everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pytc_hip.h"

namespace pytc {

typedef __bf16 bf16_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

constexpr int WAVE = 64;

// ---- last-error plumbing (thread local; the ABI returns an int status) -----------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
int tuning_get(const char* key, int dflt);
bool ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what);   // per (kernel, device) opt-in above 64 KB of dynamic LDS
bool take_launch_failure();   // a launch helper gave up before launching (error string set): PYTC_LAUNCH_CHECK turns it into a status

#define PYTC_REQUIRE(cond, ...)                     \
  do {                                              \
    if (!(cond)) {                                  \
      pytc::set_error(__VA_ARGS__);                 \
      return PYTC_ERR_INVALID;                      \
    }                                               \
  } while (0)

#define PYTC_LAUNCH_CHECK(name)                                  \
  do {                                                           \
    if (pytc::take_launch_failure()) return PYTC_ERR_HIP;        \
    hipError_t e_ = hipGetLastError();                           \
    if (e_ != hipSuccess) return pytc::hip_fail(e_, name);       \
  } while (0)

// ---- element traits: T in {float, bf16_t}; vectors of VEC channels --------------------
template <typename T, int VEC>
struct VecIO;

template <int VEC>
struct VecIO<float, VEC> {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  static __device__ __forceinline__ void load(const float* p, float (&v)[VEC]) {
    vec_t t = *reinterpret_cast<const vec_t*>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = t[i];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[VEC]) {
    vec_t t;
#pragma unroll
    for (int i = 0; i < VEC; ++i) t[i] = v[i];
    *reinterpret_cast<vec_t*>(p) = t;
  }
};
template <>
struct VecIO<float, 1> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};

template <int VEC>
struct VecIO<bf16_t, VEC> {
  typedef __bf16 vec_t __attribute__((ext_vector_type(VEC)));
  typedef float fvec_t __attribute__((ext_vector_type(VEC)));
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[VEC]) {
    vec_t t = *reinterpret_cast<const vec_t*>(p);
    fvec_t f = __builtin_convertvector(t, fvec_t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = f[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[VEC]) {
    fvec_t f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = v[i];
    *reinterpret_cast<vec_t*>(p) = __builtin_convertvector(f, vec_t);
  }
};
template <>
struct VecIO<bf16_t, 1> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[1]) { v[0] = (float)*p; }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[1]) { *p = (bf16_t)v[0]; }
};

template <typename T>
__device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v) { return (T)v; }

// ---- wave64 reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// erf GELU (torch.nn.functional.gelu(approximate='none')) with erf from Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7), evaluated as 1+erf(z) = p(t)exp(-z^2) for z<0 and 2 - p(t)exp(-z^2) for
// z>0 so the negative tail has no cancellation.  ~14 VALU + v_rcp_f32 + v_exp_f32, branch free.
__device__ __forceinline__ float gelu_erf(float x) {
  const float az = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float pe = p * __builtin_amdgcn_exp2f(-az * az * 1.44269504088896340736f);
  const float one_plus_erf = x < 0.f ? pe : 2.0f - pe;
  return 0.5f * x * one_plus_erf;
}

// d/dx gelu(x) = Phi(x) + x*phi(x) with the same A&S erf: exp(-x^2/2) serves both the erf tail and the pdf.
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float az = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-az * az * 1.44269504088896340736f);     // exp(-x^2/2)
  const float pe = p * e;
  const float one_plus_erf = x < 0.f ? pe : 2.0f - pe;
  return fmaf(x * 0.3989422804014327f, e, 0.5f * one_plus_erf);
}

// Sigmoid-form GELU for the bf16 fast path:  x * sigmoid(x * (a + b x^2 + c x^4)), minimax fit of the erf
// GELU on [-8, 8] (max abs error 2.5e-5, i.e. far below the bf16 rounding of the activation it feeds);
// x^2 is clamped at 64 so the odd polynomial keeps its sign outside the fitted range (sigmoid is saturated
// there).  7 VALU + v_exp_f32 + v_rcp_f32, about half the cost of gelu_erf.
__device__ __forceinline__ float gelu_fast(float x) {
  const float x2 = fminf(x * x, 64.0f);
  // coefficients pre-multiplied by -log2(e)
  float p = fmaf(x2, 1.0142630e-3f, -1.0677572e-1f);
  p = fmaf(p, x2, -2.3011213f);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// gelu_fast AND its derivative from one exponential and one reciprocal (round 6, mixer_bwd_rc_kernel: that kernel is bound by VALU issue,
// and gelu_fast + gelu_erf_grad cost 21 VALU + 4 quarter-rate transcendentals per element).  With u = x p(x^2), e = 2^u, s = 1 / (1 + e):
//     g = x s  (the bits of gelu_fast),     g' = s + x s (1 - s) (-ln 2) du/dx,     du/dx = p + 2 x^2 p'(x^2)
// the derivative OF THE FUNCTION THE FORWARD EVALUATED; against the exact erf form |g' - gelu'| <= 1.1e-4 over the whole axis (max at
// |x| = 0.93; bf16 rounding of g' ~ 1 is 2e-3).  1 - s, not e s: e = inf, s = 0 for x < -26 would give NaN.  15 VALU + 2 transcendentals.
__device__ __forceinline__ void gelu_fast_with_grad(float x, float& g, float& gd) {
  const float x2 = fminf(x * x, 64.0f);
  float p = fmaf(x2, 1.0142630e-3f, -1.0677572e-1f);
  p = fmaf(p, x2, -2.3011213f);
  const float e = __builtin_amdgcn_exp2f(x * p);
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  g = x * s;
  const float q = fmaf(x2, 2.0f * 1.0142630e-3f, -1.0677572e-1f);
  const float du = fmaf(x2 + x2, q, p);
  gd = fmaf(g * (1.0f - s), du * -0.69314718055994530942f, s);
}

// Packed-fp16 GELU for the fused mixers (round 2).  SQ counters put the bf16 mixers at ~70 % VALU busy with v_exp_f32 +
// v_rcp_f32 (quarter rate) as 60 % of it; packed fp32 FMAs bring nothing on gfx950 (profiles/r02_gelu_valu_probe.txt), but
// v_pk_*_f16 really does process two values per lane and instruction.  Transcendental-free form
//     gelu(x) = max(x, 0) - r(min(|x|, U)),   r(u) = u * Phi(-u)   (a bump: 0 at 0, 0.17 at 0.75, 3.3e-4 at U = 3.75)
// with r as a degree-7 polynomial in t = 2u/U - 1 (Chebyshev fit, |fit error| <= 1.0e-4): and, min, fma, 7 fma, max, sub =
// 12 packed instructions per TWO elements (6 per element vs 7 VALU + 2 quarter-rate transcendentals = ~15 for gelu_fast).
// Emulated over every fp16 input in [-8, 8]: max |error| 1.1e-3 at |x| ~ 3 (= half an fp16 ulp of the RESULT, whose
// spacing there is 2^-9; bf16, which the result used to be rounded to, has 2^-6), mean 1.5e-4.  The result IS the f16 B
// operand of the projecting MFMA (v_mfma_f32_16x16x32_f16), so the hidden activation carries 11 mantissa bits instead of
// bf16's 8.  Inputs are converted with round-toward-zero saturation (v_cvt_pkrtz_f16_f32: finite in, finite out).
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ h2_t gelu_h2(h2_t x) {
  const h2_t U = {(_Float16)3.75f, (_Float16)3.75f};
  const h2_t u = __builtin_elementwise_min(__builtin_elementwise_abs(x), U);
  const h2_t t = u * (h2_t){(_Float16)(2.0f / 3.75f), (_Float16)(2.0f / 3.75f)} + (h2_t){(_Float16)-1.0f, (_Float16)-1.0f};
#define PYTC_H2C(v) ((h2_t){(_Float16)(v), (_Float16)(v)})
  h2_t p = PYTC_H2C(-6.166994737e-02f);
  p = p * t + PYTC_H2C(6.367329291e-02f);
  p = p * t + PYTC_H2C(1.683184914e-01f);
  p = p * t + PYTC_H2C(-3.056981228e-01f);
  p = p * t + PYTC_H2C(7.903670132e-02f);
  p = p * t + PYTC_H2C(1.852329106e-01f);
  p = p * t + PYTC_H2C(-1.855973189e-01f);
  p = p * t + PYTC_H2C(5.694380662e-02f);
#undef PYTC_H2C
  return __builtin_elementwise_max(x, (h2_t){(_Float16)0.0f, (_Float16)0.0f}) - p;
}

// eight fp32 pre-activations -> gelu -> the f16 B fragment of the next MFMA: gelu_h2 on four pairs, written step by step ACROSS the pairs.
// The polynomial is a chain of dependent v_pk_fma_f16, and hipcc pads every dependent pair of them with an `s_nop 0` when it emits one
// chain after the other (64 of the 201 instructions of a mixer's hidden-chunk loop were those nops); with each step's four independent
// packed instructions side by side the next step's operands are four instructions old and the nops are gone.  Same operations per
// element in the same order: same bits as four gelu_h2 calls.
__device__ __forceinline__ h8_t gelu_h8_from_f32(const float (&v)[8]) {
#define PYTC_H2C(c) ((h2_t){(_Float16)(c), (_Float16)(c)})
  h2_t x[4], t[4], p[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]));
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = __builtin_elementwise_min(__builtin_elementwise_abs(x[q]), PYTC_H2C(3.75f));
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = t[q] * PYTC_H2C(2.0f / 3.75f) + PYTC_H2C(-1.0f);
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = PYTC_H2C(-6.166994737e-02f) * t[q] + PYTC_H2C(6.367329291e-02f);
#define PYTC_STEP(c)                                                    \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) p[q] = p[q] * t[q] + PYTC_H2C(c);
  PYTC_STEP(1.683184914e-01f)
  PYTC_STEP(-3.056981228e-01f)
  PYTC_STEP(7.903670132e-02f)
  PYTC_STEP(1.852329106e-01f)
  PYTC_STEP(-1.855973189e-01f)
  PYTC_STEP(5.694380662e-02f)
#undef PYTC_STEP
  h8_t out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const h2_t g = __builtin_elementwise_max(x[q], PYTC_H2C(0.0f)) - p[q];
    out[2 * q] = g[0];
    out[2 * q + 1] = g[1];
  }
#undef PYTC_H2C
  return out;
}

// Bijective XCD-aware remap of a 1-D block index: hardware places block b on XCD b % 8, so logical
// neighbours (which share halos / operand panels) are given to the SAME XCD's L2, in dispatch order.
__device__ __forceinline__ int xcd_swizzle(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace pytc
