// VALU issue-rate probe (round 4): SIMD cycles per wave64 instruction for the instruction classes the mixers and the z-march
// depthwise conv are made of, at 1 / 2 / 4 / 8 waves per SIMD.  8 independent dependency chains per wave, 16 instructions per asm
// block, s_memtime around the loop of wave 0 (shader clock ticks) and wall time for the chip-wide rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/issue_rate_probe tools/probes/issue_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define REP8(OP, A, X, W)                                                                                                  \
  asm volatile(OP " %0, %8, %9, %0\n " OP " %1, %8, %9, %1\n " OP " %2, %8, %9, %2\n " OP " %3, %8, %9, %3\n " OP            \
                  " %4, %8, %9, %4\n " OP " %5, %8, %9, %5\n " OP " %6, %8, %9, %6\n " OP " %7, %8, %9, %7\n"                \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7])              \
               : "v"(X), "v"(W))
#define REP8_2(OP, A, X)                                                                                                   \
  asm volatile(OP " %0, %8, %0\n " OP " %1, %8, %1\n " OP " %2, %8, %2\n " OP " %3, %8, %3\n " OP " %4, %8, %4\n " OP       \
                  " %5, %8, %5\n " OP " %6, %8, %6\n " OP " %7, %8, %7\n"                                                   \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7])              \
               : "v"(X))

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, long* ticks, int iters) {
  float a[8];
  f2 p[8];
  h2 q[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { a[i] = 1.f + i; p[i] = f2{1.f + i, 2.f}; q[i] = h2{(_Float16)(1.f + i), (_Float16)0.5f}; u[i] = 0x3f800000u + i; }
  float x = 0.5f + threadIdx.x * 1e-9f, w = 1.0001f;
  f2 px = {x, x}, pw = {w, w};
  h2 qx = {(_Float16)0.5f, (_Float16)0.25f}, qw = {(_Float16)1.001f, (_Float16)0.999f};
  unsigned ux = 0xffff0000u;
  long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (MODE == 0) REP8("v_fma_f32", a, x, w);
      if (MODE == 1) REP8("v_pk_fma_f32", p, px, pw);
      if (MODE == 2) REP8("v_pk_fma_f16", q, qx, qw);
      if (MODE == 3) REP8_2("v_and_b32", u, ux);
      if (MODE == 4) REP8_2("v_pk_max_f16", q, qx);
      if (MODE == 5) REP8_2("v_cvt_pkrtz_f16_f32", a, x);      // dst = pack(cvt(x), cvt(dst)): chain on dst
      if (MODE == 6) REP8_2("v_cvt_pk_bf16_f32", a, x);
      if (MODE == 7) REP8_2("v_exp_f32", a, x);                 // 2-operand form unused; see below
      if (MODE == 8) REP8_2("v_lshlrev_b32", u, ux);
      if (MODE == 9) REP8("v_fma_mix_f32", a, qx, w);
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)q[i][0] + (float)q[i][1] + __uint_as_float(u[i]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE> void run(const char* name, int waves_per_simd, float* d, long* dt) {
  const int iters = 4000, blocks = 256 * waves_per_simd;       // 256-thread block = one wave per SIMD of a CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dt, iters);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dt, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long ticks; hipMemcpy(&ticks, dt, 8, hipMemcpyDeviceToHost);
  const double per_wave = (double)iters * 16;
  // s_memtime ticks at a fixed 100 MHz on gfx9: use the wall time and an assumed 2.4 GHz shader clock for the SIMD-cycle figure
  const double simd_cycles = ms * 1e-3 * 2.4e9 / (per_wave * waves_per_simd);
  printf("%-22s waves/SIMD=%d  %.3f ms  %6.2f SIMD-cycles per wave-instr (at 2.4 GHz)   memtime ticks %ld\n", name, waves_per_simd, ms,
         simd_cycles, ticks);
}
int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  long* dt; hipMalloc(&dt, 8);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w, d, dt); run<1>("v_pk_fma_f32", w, d, dt); run<2>("v_pk_fma_f16", w, d, dt); run<3>("v_and_b32", w, d, dt);
    run<4>("v_pk_max_f16", w, d, dt); run<5>("v_cvt_pkrtz_f16_f32", w, d, dt); run<6>("v_cvt_pk_bf16_f32", w, d, dt);
    run<8>("v_lshlrev_b32", w, d, dt); run<9>("v_fma_mix_f32", w, d, dt);
  }
  return 0;
}
