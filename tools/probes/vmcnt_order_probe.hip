// Do global loads retire in order on gfx950?  Each wave issues a cold load A, then a hot load B (L1/L2 resident), waits
// vmcnt(1) (= "everything but the youngest"), and checks that A's destination already holds the loaded value.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* cold, const unsigned* hot, unsigned* bad, unsigned* seen, long n_cold, int iters) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned warm = hot[threadIdx.x];                       // make B hot
  unsigned nbad = 0, acc = warm;
  for (int it = 0; it < iters; ++it) {
    const long idx = (tid * 7919 + (long)it * 104729 * 64) % n_cold;
    const unsigned* pa = cold + idx;
    const unsigned* pb = hot + threadIdx.x;
    unsigned a = 0xdeadbeefu, b;
    asm volatile("global_load_dword %0, %2, off\n global_load_dword %1, %3, off\n s_waitcnt vmcnt(1)\n"
                 : "+v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
    const unsigned got = a;                               // must be cold[idx] if loads retire in order
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
    if (got != (unsigned)(idx * 2654435761u)) ++nbad;
    acc += b + a;
  }
  atomicAdd(bad, nbad);
  if (acc == 0x12345u) seen[0] = acc;
}
__global__ void fill(unsigned* p, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) p[i] = (unsigned)(i * 2654435761u);
}
int main() {
  const long n = 1L << 28;                                // 1 GiB of cold data
  unsigned *cold, *hot, *bad, *seen;
  hipMalloc(&cold, n * 4); hipMalloc(&hot, 4096); hipMalloc(&bad, 4); hipMalloc(&seen, 4);
  hipMemset(bad, 0, 4); hipMemset(hot, 1, 4096);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, n);
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, cold, hot, bad, seen, n, 200);
  unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("lanes that saw a stale A after vmcnt(1): %u of %ld\n", h, 2048L * 256 * 200);
  return 0;
}
