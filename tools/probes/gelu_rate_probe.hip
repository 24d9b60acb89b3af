// VALU cost of GELU variants and of the primitives behind them on gfx950: ns per element with 4 waves / SIMD, all operands in
// registers.  Settles (round 2) whether a transcendental-free, packed-fp32 polynomial GELU beats the sigmoid form
// (7 VALU + v_exp_f32 + v_rcp_f32 per element) the fused mixers use.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gelu_rate_probe.hip -o /tmp/gelu_probe && /tmp/gelu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gelu_fast(float x) {
  const float x2 = fminf(x * x, 64.0f);
  float p = fmaf(x2, 1.0142630e-3f, -1.0677572e-1f);
  p = fmaf(p, x2, -2.3011213f);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// gelu(x) = max(x, 0) - r(min(|x|, 4.5)),  r(u) = u * Phi(-u): degree-10 minimax in t = 2u/4.5 - 1 (|err| <= 1.5e-5 in fp32)
__device__ __forceinline__ f2 gelu_poly2(f2 x) {
  const float C[11] = {2.749713780e-02f, -1.330395468e-01f, 2.465921861e-01f, -1.472151196e-01f, -2.029664835e-01f,
                       4.347783315e-01f, -2.049124982e-01f, -1.763101730e-01f, 1.763865979e-01f, 2.177998023e-02f,
                       -4.258929477e-02f};
  f2 u = {fminf(fabsf(x[0]), 4.5f), fminf(fabsf(x[1]), 4.5f)};
  f2 t = u * (f2){2.0f / 4.5f, 2.0f / 4.5f} + (f2){-1.f, -1.f};
  f2 p = {C[10], C[10]};
#pragma unroll
  for (int k = 9; k >= 0; --k) p = p * t + (f2){C[k], C[k]};
  f2 m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return m - p;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f2 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) { v[i][0] = gelu_fast(v[i][0]) + 0.5f; v[i][1] = gelu_fast(v[i][1]) + 0.5f; }
      else if (MODE == 1) { v[i] = gelu_poly2(v[i]) + (f2){0.5f, 0.5f}; }
      else if (MODE == 2) { v[i][0] = __builtin_amdgcn_exp2f(v[i][0]); v[i][1] = __builtin_amdgcn_exp2f(v[i][1]); }
      else if (MODE == 3) { v[i][0] = __builtin_amdgcn_rcpf(v[i][0]); v[i][1] = __builtin_amdgcn_rcpf(v[i][1]); }
      else if (MODE == 4) { v[i][0] = fmaf(v[i][0], 1.0001f, 0.5f); v[i][1] = fmaf(v[i][1], 1.0001f, 0.5f); }
      else if (MODE == 5) { v[i] = v[i] * (f2){1.0001f, 1.0001f} + (f2){0.5f, 0.5f}; }
      else if (MODE == 6) { v[i][0] = fminf(fabsf(v[i][0]), 4.5f); v[i][1] = fmaxf(v[i][1], 0.f); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* d, int per_elem_ops) {
  const int iters = 4000, blocks = 256 * 4 * 4;     // 4 waves / SIMD on 256 CUs
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double elems = (double)blocks * 256 * iters * 8;
  // cycles per wave-element on one SIMD: elems / (1024 SIMDs) / 64 lanes = wave-elements per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / (elems / 64.0 / 1024.0);
  printf("%-28s %8.3f ms  %7.2f ps/elem   ~%6.2f SIMD-cycles per wave64-element (at 2.4 GHz)\n", name, ms, ms * 1e9 / elems, cyc);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 4 * 256 * 4);
  run<4>("v_fma_f32", d, 1);
  run<5>("v_pk_fma_f32 (2 elems)", d, 1);
  run<6>("v_min|abs| / v_max", d, 1);
  run<2>("v_exp_f32", d, 1);
  run<3>("v_rcp_f32", d, 1);
  run<0>("gelu_fast (exp+rcp)", d, 1);
  run<1>("gelu_poly2 (packed, deg 10)", d, 1);
  return 0;
}
