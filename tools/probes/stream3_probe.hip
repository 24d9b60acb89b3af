// Round 5: what does HBM give a THREE-STREAM kernel (read t, read x, write y: the level-0 mixer's traffic, 64 B per voxel per stream)?
// The level-0 mixer moves 2.16 GB in 428 us = 5.05 TB/s, torch's copy_ 5.3 TB/s, a fill 6.8 TB/s.  Variants of y = t + x (bf16x8 per lane):
//   cache policy of the loads / stores (plain, nontemporal), rows per wave (contiguous bytes per wave), block -> address maps (linear,
//   XCD-contiguous: the blocks of one XCD walk one eighth of the tensor), persistent grid-stride.
//     hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/stream3_probe tools/probes/stream3_probe.hip && tools/probes/bin/stream3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NTL, bool NTS>
__device__ __forceinline__ void body(const u32x4* __restrict__ t, const u32x4* __restrict__ x, u32x4* __restrict__ y, long i) {
  u32x4 a, b;
  if (NTL) { a = __builtin_nontemporal_load(t + i); b = __builtin_nontemporal_load(x + i); }
  else { a = t[i]; b = x[i]; }
  u32x4 c;
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = a[k] + b[k];
  if (NTS) __builtin_nontemporal_store(c, y + i); else y[i] = c;
}

// MAP 0: linear (block b owns chunk b).  MAP 1: XCD-contiguous (block b -> chunk (b % 8) * (nb / 8) + b / 8).  MAP 2: persistent grid-stride.
template <bool NTL, bool NTS, int PER, int MAP>
__global__ void __launch_bounds__(256) stream3(const u32x4* __restrict__ t, const u32x4* __restrict__ x, u32x4* __restrict__ y, long n16) {
  long nb = gridDim.x;
  long b = blockIdx.x;
  if (MAP == 1) { const long q = nb >> 3; b = (b & 7) * q + (b >> 3); if (b >= nb) b = blockIdx.x; }
  if (MAP == 2) {
    for (long base = b * 256 * PER; base < n16; base += nb * 256 * PER) {
#pragma unroll
      for (int p = 0; p < PER; ++p) { const long i = base + p * 256 + threadIdx.x; if (i < n16) body<NTL, NTS>(t, x, y, i); }
    }
    return;
  }
  const long base = b * 256 * PER;
  u32x4 a[PER], c[PER];
#pragma unroll
  for (int p = 0; p < PER; ++p) { const long i = base + p * 256 + threadIdx.x; if (i < n16) { a[p] = NTL ? __builtin_nontemporal_load(t + i) : t[i]; c[p] = NTL ? __builtin_nontemporal_load(x + i) : x[i]; } }
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const long i = base + p * 256 + threadIdx.x;
    if (i < n16) {
      u32x4 r;
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = a[p][k] + c[p][k];
      if (NTS) __builtin_nontemporal_store(r, y + i); else y[i] = r;
    }
  }
}

template <bool NTS> __global__ void __launch_bounds__(256) fill(u32x4* __restrict__ y, long n16) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) { u32x4 v = {1u, 2u, 3u, 4u}; if (NTS) __builtin_nontemporal_store(v, y + i); else y[i] = v; }
}
template <bool NTL> __global__ void __launch_bounds__(256) readsum(const u32x4* __restrict__ t, unsigned* __restrict__ out, long n16) {
  const long base = (long)blockIdx.x * 256 * 4;
  unsigned s = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) { const long i = base + p * 256 + threadIdx.x; if (i < n16) { u32x4 a = NTL ? __builtin_nontemporal_load(t + i) : t[i]; s += a[0] ^ a[1] ^ a[2] ^ a[3]; } }
  if (s == 0x12345677u) out[0] = s;
}

template <typename F>
static float time_us(F launch, int reps = 20) {
  for (int i = 0; i < 4; ++i) launch();
  hipEvent_t s, e; CHECK(hipEventCreate(&s)); CHECK(hipEventCreate(&e));
  CHECK(hipEventRecord(s));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e)); CHECK(hipEventSynchronize(e));
  float ms; CHECK(hipEventElapsedTime(&ms, s, e));
  return ms / reps * 1e3f;
}

int main() {
  const long bytes = 8L * 112 * 112 * 112 * 32 * 2;       // one level-0 tensor of an 8-window batch
  const long n16 = bytes / 16;
  u32x4 *t, *x, *y; unsigned* out;
  CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&x, bytes)); CHECK(hipMalloc(&y, bytes)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(t, 1, bytes)); CHECK(hipMemset(x, 2, bytes));
  printf("three streams of %.3f GB (read t, read x, write y)\n", bytes / 1e9);
  auto rep = [&](const char* name, float us, int streams) { printf("  %-58s %8.1f us  %6.0f GB/s\n", name, us, streams * bytes / us / 1e3); };
#define RUN(NTL, NTS, PER, MAP, label) { const long nb = MAP == 2 ? 256 * 8 : (n16 + 256 * PER - 1) / (256 * PER); \
    rep(label, time_us([&] { hipLaunchKernelGGL((stream3<NTL, NTS, PER, MAP>), dim3((unsigned)nb), dim3(256), 0, 0, t, x, y, n16); }), 3); }
  RUN(false, false, 1, 0, "plain loads / stores, 1 x 16 B per lane, linear");
  RUN(false, false, 4, 0, "plain, 4 x 16 B per lane (16 KB per block per stream)");
  RUN(true, false, 4, 0, "nt loads, plain stores, 4 per lane");
  RUN(false, true, 4, 0, "plain loads, nt stores, 4 per lane");
  RUN(true, true, 4, 0, "nt loads + nt stores, 4 per lane");
  RUN(true, true, 1, 0, "nt loads + nt stores, 1 per lane");
  RUN(true, true, 8, 0, "nt loads + nt stores, 8 per lane");
  RUN(false, false, 4, 1, "plain, 4 per lane, XCD-contiguous block map");
  RUN(true, true, 4, 1, "nt + nt, 4 per lane, XCD-contiguous block map");
  RUN(false, false, 4, 2, "plain, persistent grid-stride (2048 blocks)");
  RUN(true, true, 4, 2, "nt + nt, persistent grid-stride (2048 blocks)");
  rep("fill, plain stores", time_us([&] { hipLaunchKernelGGL(fill<false>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, y, n16); }), 1);
  rep("fill, nt stores", time_us([&] { hipLaunchKernelGGL(fill<true>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, y, n16); }), 1);
  rep("read, plain loads (4 per lane)", time_us([&] { hipLaunchKernelGGL(readsum<false>, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, t, out, n16); }), 1);
  rep("read, nt loads (4 per lane)", time_us([&] { hipLaunchKernelGGL(readsum<true>, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, t, out, n16); }), 1);
  rep("hipMemcpyDtoD (2 streams)", time_us([&] { CHECK(hipMemcpyAsync(y, t, bytes, hipMemcpyDeviceToDevice, 0)); }), 2);
  return 0;
}
