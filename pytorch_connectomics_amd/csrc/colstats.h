// Per-(sample, channel) column sums over the voxel rows of an NDHWC tensor, 16 bytes per lane per load:
//   MODE 0: (sum x, sum x^2)                      -> GroupNorm / InstanceNorm statistics
//   MODE 1: (sum d, sum d * xhat), xhat=(t-mean)*rstd -> GroupNorm backward statistics
// Workgroup = one (row slot, sample); lane = (16-byte channel chunk, row lane); partials [N][slots][2][C] are reduced
// later in slot order, so the result does not depend on the launch.  Needs C % (16/sizeof(T)) == 0.
#pragma once
#include "pytc_common.h"

namespace pytc {

static inline int colstats_slots(long rows) {
  const long s = rows / 64;
  return (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256)
colstats_kernel(const T* __restrict__ a, const T* __restrict__ t, const float* __restrict__ mr,
                float* __restrict__ stats, long rows, int C, int slots, long rows_per_slot) {
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ float lds[2 * 256 * EPV];                 // [row lane][2][Cw * EPV]
  const int n = blockIdx.y, slot = blockIdx.x;
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
  const T* an = a + (long)n * rows * C;
  const T* tn = MODE == 1 ? t + (long)n * rows * C : nullptr;
  const int chunks = C / EPV;
  for (int k0 = 0; k0 < chunks; k0 += 256) {
    const int Cw = (chunks - k0) < 256 ? (chunks - k0) : 256;
    const int RL = 256 / Cw;
    const int ck = threadIdx.x % Cw, rl = threadIdx.x / Cw;
    const int c = (k0 + ck) * EPV;
    if (rl < RL) {
      float s1[EPV], s2[EPV], mean[EPV], rstd[EPV];
#pragma unroll
      for (int i = 0; i < EPV; ++i) {
        s1[i] = 0.f; s2[i] = 0.f;
        mean[i] = MODE == 1 ? mr[((long)n * 2 + 0) * C + c + i] : 0.f;
        rstd[i] = MODE == 1 ? mr[((long)n * 2 + 1) * C + c + i] : 0.f;
      }
#pragma unroll 4
      for (long r = r0 + rl; r < r1; r += RL) {
        float v[EPV];
        VecIO<T, EPV>::load(an + r * C + c, v);
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < EPV; ++i) { s1[i] += v[i]; s2[i] = fmaf(v[i], v[i], s2[i]); }
        } else {
          float u[EPV];
          VecIO<T, EPV>::load(tn + r * C + c, u);
#pragma unroll
          for (int i = 0; i < EPV; ++i) { s1[i] += v[i]; s2[i] = fmaf(v[i], (u[i] - mean[i]) * rstd[i], s2[i]); }
        }
      }
#pragma unroll
      for (int i = 0; i < EPV; ++i) {
        lds[((rl * 2 + 0) * Cw + ck) * EPV + i] = s1[i];
        lds[((rl * 2 + 1) * Cw + ck) * EPV + i] = s2[i];
      }
    }
    __syncthreads();
    const int width = Cw * EPV;
    for (int i = threadIdx.x; i < 2 * width; i += 256) {
      const int which = i / width, e = i % width;
      float acc = 0.f;
      for (int q = 0; q < RL; ++q) acc += lds[(q * 2 + which) * width + e];
      stats[(((long)n * slots + slot) * 2 + which) * C + k0 * EPV + e] = acc;
    }
    __syncthreads();
  }
}

}  // namespace pytc
