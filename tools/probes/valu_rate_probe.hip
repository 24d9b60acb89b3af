// VALU issue-rate probe: cycles per wave64 instruction for v_fma_f32 vs v_pk_fma_f32 (and a ds_read_b64-fed mix).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f2 a0 = {1.f, 2.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  f2 x = {0.5f, 0.25f}, w = {1.0001f, 0.9999f};
  x[0] += threadIdx.x * 1e-9f;
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {   // 16 v_fma_f32
        asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                     "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                     : "+v"(a0[0]), "+v"(a1[0]), "+v"(a2[0]), "+v"(a3[0]), "+v"(a4[0]), "+v"(a5[0]), "+v"(a6[0]), "+v"(a7[0]) : "v"(x[0]), "v"(w[0]));
        asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                     "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                     : "+v"(a0[1]), "+v"(a1[1]), "+v"(a2[1]), "+v"(a3[1]), "+v"(a4[1]), "+v"(a5[1]), "+v"(a6[1]), "+v"(a7[1]) : "v"(x[1]), "v"(w[1]));
      } else {           // 8 v_pk_fma_f32 (same flops)
        asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                     "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
      }
    }
  }
  long t1 = __builtin_readcyclecounter();
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int MODE> void run(const char* name, int blocks, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  const double ninstr = (double)iters * 8 * (MODE == 0 ? 16 : 8);
  printf("%-14s blocks=%5d  %.3f ms  wave0: %.2f clk-ticks/instr  (flops/wave-instr=%d)  => %.1f TFLOP/s\n", name, blocks, ms,
         cyc / ninstr, MODE == 0 ? 128 : 256, (double)blocks * 4 * iters * 8 * 16 * 128 / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  for (int blocks : {256, 512, 1024, 2048}) { run<0>("v_fma_f32", blocks, d); run<1>("v_pk_fma_f32", blocks, d); }
  return 0;
}
