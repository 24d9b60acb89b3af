// Concurrency probe for packed-fp32 VALU forms on gfx950 (round 3).
// Observation that led here: dwconv3d_xblock_kernel (compiler-generated v_pk_fma_f32 with `op_sel:[0,1,0]`, i.e. the low
// result lane reading the HIGH dword of src1) produced wrong values in one quarter-wave of one register, but only while an
// MFMA kernel of ANOTHER HIP stream shared the GPU; the same kernel without packed-fp32 code is exact.
// This probe runs a victim that chains one packed-fp32 FMA form (plain | op_sel_hi:[1,0,1] | op_sel:[0,1,0]) -- optionally
// with its src1 freshly returned from LDS, as in the depthwise kernel -- on one stream while an MFMA loop runs on a second
// stream, and counts lanes whose result differs from the quiet run of the same kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int FORM, bool LDS>
__global__ void __launch_bounds__(256) victim(float* out, int iters) {
  __shared__ f4 taps[256];
  const int t = threadIdx.x;
  taps[t] = f4{1.0f + t * 1e-3f, 0.5f + t * 2e-3f, 0.25f + t * 3e-3f, 0.125f + t * 5e-4f};
  __syncthreads();
  f2 acc[6];
  for (int j = 0; j < 6; ++j) acc[j] = f2{0.001f * (t + j), 0.002f * (t + 3 * j)};
  f2 x = {0.75f + 1e-4f * t, 0.3f - 2e-4f * t};
  for (int i = 0; i < iters; ++i) {
    f4 w4 = LDS ? taps[(t + i) & 255] : f4{1.0f + i * 1e-6f, 0.5f, 0.25f, 0.125f};
    f2 w = {w4[0], w4[1]};
    f2 w2 = {w4[2], w4[3]};
#define PK(A, X, W, SUFFIX) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 " SUFFIX : "+v"(A) : "v"(X), "v"(W))
    if (FORM == 0) { PK(acc[0], x, w, ""); PK(acc[1], x, w2, ""); PK(acc[2], x, w, ""); PK(acc[3], x, w2, ""); PK(acc[4], x, w, ""); PK(acc[5], x, w2, ""); }
    if (FORM == 1) { PK(acc[0], x, w, "op_sel_hi:[1,0,1]"); PK(acc[1], x, w2, "op_sel_hi:[1,0,1]"); PK(acc[2], x, w, "op_sel_hi:[1,0,1]");
                     PK(acc[3], x, w2, "op_sel_hi:[1,0,1]"); PK(acc[4], x, w, "op_sel_hi:[1,0,1]"); PK(acc[5], x, w2, "op_sel_hi:[1,0,1]"); }
    if (FORM == 2) { PK(acc[0], x, w, "op_sel:[0,1,0]"); PK(acc[1], x, w2, "op_sel:[0,1,0]"); PK(acc[2], x, w, "op_sel:[0,1,0]");
                     PK(acc[3], x, w2, "op_sel:[0,1,0]"); PK(acc[4], x, w, "op_sel:[0,1,0]"); PK(acc[5], x, w2, "op_sel:[0,1,0]"); }
    if (FORM == 3) {   // the mix the compiler emitted: alternate the two selected forms
      PK(acc[0], x, w, "op_sel_hi:[1,0,1]"); PK(acc[1], x, w, "op_sel:[0,1,0]"); PK(acc[2], x, w2, "op_sel_hi:[1,0,1]");
      PK(acc[3], x, w, "op_sel_hi:[1,0,1]"); PK(acc[4], x, w, "op_sel:[0,1,0]"); PK(acc[5], x, w2, "op_sel_hi:[1,0,1]"); }
#undef PK
    if (FORM == 4 || FORM == 5) {
      // src1 (w) is the target of a ds_read_b64 issued immediately after the packed FMA that reads it
      const unsigned addr = (unsigned)(((t + i + 1) & 255) * 16);
      f2 wn = w;
      if (FORM == 4) asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[0,1,0]\n ds_read_b64 %1, %3\n s_waitcnt lgkmcnt(0)" : "+v"(acc[0]), "+v"(wn) : "v"(x), "v"(addr) : "memory");
      else asm volatile("v_pk_fma_f32 %0, %2, %1, %0\n ds_read_b64 %1, %3\n s_waitcnt lgkmcnt(0)" : "+v"(acc[0]), "+v"(wn) : "v"(x), "v"(addr) : "memory");
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(x), "v"(wn));
    }
    for (int j = 0; j < 6; ++j) acc[j] = acc[j] * 0.5f;      // keep the values bounded (plain v_pk_mul / v_mul)
  }
  float* o = out + ((long)blockIdx.x * 256 + t) * 12;
  for (int j = 0; j < 6; ++j) { o[2 * j] = acc[j][0]; o[2 * j + 1] = acc[j][1]; }
}

__global__ void __launch_bounds__(256) aggressor_mfma(float* out, int iters) {
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * j); }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void __launch_bounds__(256) aggressor_valu(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters * 16; ++i) { a = __builtin_fmaf(a, b, c); c = __builtin_fmaf(c, 0.999f, a * 1e-6f); }
  out[blockIdx.x * 256 + threadIdx.x] = a + c;
}

template <int FORM, bool LDS>
static void trial(const char* name, int aggr, hipStream_t sa, hipStream_t sv, float* dout, float* dag) {
  const int blocks = 1024, iters = 4000, n = blocks * 256 * 12;
  std::vector<float> quiet(n), conc(n);
  hipLaunchKernelGGL((victim<FORM, LDS>), dim3(blocks), dim3(256), 0, sv, dout, iters);
  hipStreamSynchronize(sv);
  hipMemcpy(quiet.data(), dout, n * 4, hipMemcpyDeviceToHost);
  long bad = 0, runs = 0;
  for (int rep = 0; rep < 10; ++rep) {
    if (aggr == 1) hipLaunchKernelGGL(aggressor_mfma, dim3(2048), dim3(256), 0, sa, dag, 20000);
    if (aggr == 2) hipLaunchKernelGGL(aggressor_valu, dim3(2048), dim3(256), 0, sa, dag, 4000);
    hipLaunchKernelGGL((victim<FORM, LDS>), dim3(blocks), dim3(256), 0, sv, dout, iters);
    hipStreamSynchronize(sv);
    hipMemcpy(conc.data(), dout, n * 4, hipMemcpyDeviceToHost);
    hipStreamSynchronize(sa);
    for (int i = 0; i < n; ++i) bad += memcmp(&quiet[i], &conc[i], 4) != 0;
    ++runs;
  }
  printf("%-34s lds_fed=%d aggressor=%-5s mismatching result dwords over %ld runs: %ld\n", name, (int)LDS,
         aggr == 0 ? "none" : (aggr == 1 ? "mfma" : "valu"), runs, bad);
}

int main() {
  float *dout, *dag;
  hipMalloc(&dout, 1024L * 256 * 12 * 4);
  hipMalloc(&dag, 2048L * 256 * 4);
  hipStream_t sa, sv;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sv, hipStreamNonBlocking);
  for (int aggr = 0; aggr < 3; ++aggr) {
    trial<0, false>("v_pk_fma_f32 (no select)", aggr, sa, sv, dout, dag);
    trial<1, false>("v_pk_fma_f32 op_sel_hi:[1,0,1]", aggr, sa, sv, dout, dag);
    trial<2, false>("v_pk_fma_f32 op_sel:[0,1,0]", aggr, sa, sv, dout, dag);
    trial<3, false>("mixed (as emitted by the compiler)", aggr, sa, sv, dout, dag);
    trial<4, true>("pk_fma op_sel:[0,1,0] + ds_read over src1", aggr, sa, sv, dout, dag);
    trial<5, true>("pk_fma + ds_read over src1", aggr, sa, sv, dout, dag);
    trial<0, true>("v_pk_fma_f32 (no select)", aggr, sa, sv, dout, dag);
    trial<1, true>("v_pk_fma_f32 op_sel_hi:[1,0,1]", aggr, sa, sv, dout, dag);
    trial<2, true>("v_pk_fma_f32 op_sel:[0,1,0]", aggr, sa, sv, dout, dag);
    trial<3, true>("mixed (as emitted by the compiler)", aggr, sa, sv, dout, dag);
  }
  return 0;
}
