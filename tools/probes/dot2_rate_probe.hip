// VALU issue-rate probe (round 6): v_dot2c_f32_bf16 (2 bf16 products + fp32 accumulate per lane) against v_fma_f32 and v_pk_fma_f32.
// The depthwise weight gradient is bound by VALU issue (27 MACs per voxel and channel); its operands are bf16 in HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float a[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 1.f + i; p[i] = f2{1.f + i, 2.f + i}; }
  unsigned x = 0x3f803f80u + threadIdx.x, w = 0x3f003e80u;       // bf16 pairs
  float xf = 0.5f + threadIdx.x * 1e-9f, wf = 1.0001f;
  f2 xp = {xf, xf * 0.5f}, wp = {wf, 0.9999f};
  long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                     "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(xf), "v"(wf));
      } else if (MODE == 1) {
        asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                     "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                     : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(xp), "v"(wp));
      } else {
        asm volatile("v_dot2c_f32_bf16 %0, %8, %9\n v_dot2c_f32_bf16 %1, %8, %9\n v_dot2c_f32_bf16 %2, %8, %9\n v_dot2c_f32_bf16 %3, %8, %9\n"
                     "v_dot2c_f32_bf16 %4, %8, %9\n v_dot2c_f32_bf16 %5, %8, %9\n v_dot2c_f32_bf16 %6, %8, %9\n v_dot2c_f32_bf16 %7, %8, %9\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
      }
    }
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int MODE> void run(const char* name, int blocks, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ninstr = (double)iters * 64;
  const int macs = MODE == 0 ? 64 : 128;       // per wave instruction
  printf("%-18s blocks=%5d  %.3f ms  => %.2f T wave-instr/s chip-wide, %.1f TMAC/s\n", name, blocks, ms,
         (double)blocks * 4 * ninstr / ms / 1e9, (double)blocks * 4 * ninstr * macs / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  for (int blocks : {1024, 2048}) { run<0>("v_fma_f32", blocks, d); run<1>("v_pk_fma_f32", blocks, d); run<2>("v_dot2c_f32_bf16", blocks, d); }
  // value check: dot2c(acc = 1, a = (1.0, 1.0) bf16, b = (0.5, 0.25) bf16) = 1.75
  return 0;
}
