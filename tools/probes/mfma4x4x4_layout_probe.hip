// Probe (standalone): operand layout and issue rate of v_mfma_f32_4x4x4_16b_bf16 on gfx950 -- 16 independent 4x4x4 products per
// instruction ("blocks").  The depthwise 3x3x3 conv maps one CHANNEL to one block (csrc/dwconv_mfma_kernels.hip).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/mfma4x4x4_layout_probe.hip -o tools/probes/bin/mfma4x4x4_layout_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(const s4* a, const s4* b, f4* d) {
  f4 c = {0, 0, 0, 0};
  d[threadIdx.x] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
}

// NCH independent accumulator chains per wave, `iters` rounds: cycles per MFMA as issued by one wave / by W waves per SIMD
template <int NCH>
__global__ void rate(const s4* a, const s4* b, f4* d, int iters, long long* cyc) {
  f4 acc[NCH];
  for (int i = 0; i < NCH; ++i) acc[i] = f4{0, 0, 0, 0};
  const s4 av = a[threadIdx.x & 63], bv = b[threadIdx.x & 63];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, bv, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  f4 s = {0, 0, 0, 0};
  for (int i = 0; i < NCH; ++i) s += acc[i];
  d[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static unsigned short bf(float f) { unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

int main() {
  unsigned short ha[64 * 4], hb[64 * 4];
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
    int blk = l / 4, r = l % 4;
    ha[l * 4 + q] = bf((float)(1 + r * 4 + q + blk * 16));        // guess: lane 4b+i holds A[b][i][k = 0..3]
    hb[l * 4 + q] = bf((float)((q == 1 ? 1 : 0) * (r + 1)));      // guess: lane 4b+j holds B[b][k = 0..3][j] = (k == 1) * (j + 1)
  }
  s4 *da, *db; f4* dd; long long* dc;
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 8 << 20); (void)hipMalloc(&dc, 8);
  (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, dd);
  float hd[256]; (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
  // layout guess for D: lane 4b+j, VGPR i holds D[b][i][j] = A[b][i][1] * (j + 1)
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    int blk = l / 4, j = l % 4;
    float want = (float)((2 + i * 4 + blk * 16) * (j + 1));
    if (hd[l * 4 + i] != want) { if (bad < 8) printf("lane %d vgpr %d got %g want %g\n", l, i, hd[l * 4 + i], want); ++bad; }
  }
  printf("bf16 4x4x4 16-block layout (A: lane 4b+i, k in register; B: lane 4b+j; D: lane 4b+j, VGPR i): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  const int iters = 20000;
  for (int waves : {1, 2, 4, 8}) {
    long long c1, c8;
    rate<1><<<1, 64 * waves * 4>>>(da, db, dd, iters, dc); (void)hipMemcpy(&c1, dc, 8, hipMemcpyDeviceToHost);
    rate<8><<<1, 64 * waves * 4>>>(da, db, dd, iters, dc); (void)hipMemcpy(&c8, dc, 8, hipMemcpyDeviceToHost);
    // clock64 = s_memtime: 100 MHz constant clock on gfx9 -> convert with the shader clock afterwards; print raw too
    printf("%d waves/SIMD: dependent chain %.2f ticks/MFMA, 8 chains %.2f ticks/MFMA (per wave)\n", waves, (double)c1 / iters, (double)c8 / iters / 8);
  }
  // wall-clock rate over the whole chip: one block per CU, W waves per SIMD, 8 independent chains (and 1 dependent chain) per wave
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int W : {1, 2, 4}) {
    float ms8, ms1;
    rate<8><<<256, 256 * W>>>(da, db, dd, iters, dc);
    (void)hipEventRecord(e0);
    rate<8><<<256, 256 * W>>>(da, db, dd, iters, dc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms8, e0, e1);
    (void)hipEventRecord(e0);
    rate<1><<<256, 256 * W>>>(da, db, dd, iters, dc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms1, e0, e1);
    printf("whole chip, %d waves/SIMD: 8 chains %.2f ns per MFMA per wave = %.2f ns per MFMA per SIMD; dependent chain %.2f ns per MFMA per wave\n", W,
           ms8 * 1e6 / (iters * 8.0), ms8 * 1e6 / (iters * 8.0 * W), ms1 * 1e6 / iters);
  }
  return 0;
}
