// Probe (standalone, not part of libpytc_hip.so): depthwise 3x3x3 convolution on the matrix cores in TOEPLITZ form -- the round-4
// candidate of DESIGN.md section 7, first correct-by-construction version (index algebra: tools/history/proto_toeplitz_dwconv.py).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/toeplitz_dwconv_probe.hip -o tools/probes/bin/toeplitz_dwconv_probe
//   tools/probes/bin/toeplitz_dwconv_probe            # checks a small volume against a CPU convolution, then times 8 x 112^3 x 32
//
// One workgroup = 4 waves = one footprint of 16 rows (y) x 14 columns (x) x 32 channels, marching along a z chunk.  Per input plane:
// (a) all threads transpose the haloed NDHWC plane (18 x 16 voxels x 32 channels, 16-byte global loads) into a CHANNEL-MAJOR bf16 image
//     in LDS (ring of 3 planes);  (b) wave w owns channels 8w..8w+7, whose 5 x 8 Toeplitz A-fragments stay in registers for the whole
//     march: per channel 5 B-fragments (one 16-byte LDS read each: 8 consecutive columns of one row) and 5 v_mfma_f32_16x16x32_bf16,
//     the result (+ bias, rounded to bf16, statistics) goes to an LDS staging tile in NDHWC order;  (c) all threads store the tile with
//     16-byte writes.  Three barriers per plane: this version pins down correctness and the resource budget, not the schedule.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int TY = 16, TX = 14, WIN = 16, ROWS = TY + 2, C = 32, CPW = 8;   // rows / columns per tile, window columns, channels, per wave
constexpr int CH_PITCH = ROWS * WIN + 8;                                     // halfwords per channel image (+8: bank spread across channels)
constexpr int PLANE_HW = C * CH_PITCH;                                       // halfwords per staged plane

struct Geom { int N, D, H, W, zc, nzc, ty, tx; };

__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// PIPE = false: load a plane, unpack it, then compute (every 16-byte load is waited for on the spot: 4.5 serial HBM round trips per plane
// with one wave per SIMD -- 470 us per 2 windows, profiles/r03_toeplitz_probe_v2.txt).  PIPE = true: the loads of plane z + 2 are issued
// into registers before the MFMA phase of plane z and unpacked into the ring after it (software pipeline, +20 VGPRs).
template <bool PIPE>
__global__ void __launch_bounds__(256, 1)
toeplitz_dwconv_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, const float* __restrict__ w,
                       const float* __restrict__ bias, float* __restrict__ stats, Geom g) {
  __shared__ __attribute__((aligned(16))) unsigned short ring[3][PLANE_HW];
  __shared__ __attribute__((aligned(16))) unsigned short ostage[TY * TX * C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int fx = b % g.tx; b /= g.tx;
  const int fy = b % g.ty; b /= g.ty;
  const int zchunk = b % g.nzc;
  const int n = b / g.nzc;
  const int y0 = fy * TY, x0 = fx * TX;
  const int zs = zchunk * g.zc, ze = min(zs + g.zc, g.D);
  const long plane_elems = (long)g.H * g.W * C;
  const unsigned short* xn = x + (long)n * g.D * plane_elems;
  unsigned short* yn = y + (long)n * g.D * plane_elems;

  // ---- the 27 x 32 taps go through LDS once (coalesced), so that the 320 fragment elements per lane are unconditional LDS reads: the
  //      first version read them with 320 conditional global loads per lane, each behind its own branch and `s_waitcnt vmcnt(0)` --
  //      ~130 us of serialised round trips per workgroup, which was the whole 545 us of profiles/r03_toeplitz_probe.txt
  __shared__ float wl[27 * C];
  for (int i = tid; i < 27 * C; i += 256) wl[i] = w[i];
  __syncthreads();
  // ---- A fragments of this wave's 8 channels: afr[ch][s] = rows m (= lane & 15) of the banded Toeplitz block of tap pair s
  const int m = lane & 15, kg = lane >> 4;
  bf16x8_t afr[CPW][5];
#pragma unroll
  for (int ch = 0; ch < CPW; ++ch) {
    const int c = wave * CPW + ch;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int q = 2 * s + (kg >> 1);
      bf16x8_t f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = (kg & 1) * 8 + i, dx = j - m;
        const bool on = q <= 8 && m < TX && dx >= 0 && dx <= 2;                            // taps [kz][ky][kx][c], q = kz * 3 + ky
        const float v = wl[on ? (q * 3 + dx) * C + c : c];
        f[i] = (__bf16)(on ? v : 0.f);
      }
      afr[ch][s] = f;
    }
  }
  float bv[CPW], s1[CPW], s2[CPW];
#pragma unroll
  for (int ch = 0; ch < CPW; ++ch) { bv[ch] = bias ? bias[wave * CPW + ch] : 0.f; s1[ch] = 0.f; s2[ch] = 0.f; }

  // ---- staging of one haloed plane: 18 x 16 voxels x 4 chunks of 8 channels, transposed to [c][row][col]
  auto stage = [&](int gz, int slot) {
    unsigned short* dst = ring[slot];
    const bool zok = gz >= 0 && gz < g.D;
    for (int chunk = tid; chunk < ROWS * WIN * (C / 8); chunk += 256) {
      const int part = chunk & 3, vox = chunk >> 2;
      const int row = vox / WIN, col = vox % WIN;
      const int gy = y0 - 1 + row, gx = x0 - 1 + col;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (zok && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
        v = *reinterpret_cast<const u32x4_t*>(xn + (long)gz * plane_elems + ((long)gy * g.W + gx) * C + part * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned int d = v[i >> 1];
        dst[(part * 8 + i) * CH_PITCH + row * WIN + col] = (unsigned short)((i & 1) ? (d >> 16) : (d & 0xFFFFu));
      }
    }
  };

  // ---- the same staging split in two: `issue` puts the 16-byte loads of a plane in flight (registers), `commit` unpacks them into a slot
  constexpr int NPRE = (ROWS * WIN * (C / 8) + 255) / 256;                                      // 5 loads per thread and plane
  u32x4_t pre[NPRE];
  auto issue = [&](int gz) {
    const bool zok = gz >= 0 && gz < g.D;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int chunk = tid + k * 256;
      const int part = chunk & 3, vox = chunk >> 2;
      const int row = vox / WIN, col = vox % WIN;
      const int gy = y0 - 1 + row, gx = x0 - 1 + col;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (chunk < ROWS * WIN * (C / 8) && zok && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
        v = *reinterpret_cast<const u32x4_t*>(xn + (long)gz * plane_elems + ((long)gy * g.W + gx) * C + part * 8);
      pre[k] = v;
    }
  };
  auto commit = [&](int slot) {
    unsigned short* dst = ring[slot];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int chunk = tid + k * 256;
      if (chunk < ROWS * WIN * (C / 8)) {
        const int part = chunk & 3, vox = chunk >> 2;
        const int row = vox / WIN, col = vox % WIN;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned int d = pre[k][i >> 1];
          dst[(part * 8 + i) * CH_PITCH + row * WIN + col] = (unsigned short)((i & 1) ? (d >> 16) : (d & 0xFFFFu));
        }
      }
    }
  };

  if (PIPE) {
    issue(zs - 1); commit((zs - 1 + 3) % 3);
    issue(zs); commit(zs % 3);
    issue(zs + 1);                                     // in flight until the top of the first iteration
  } else {
    stage(zs - 1, (zs - 1 + 3) % 3);
    stage(zs, zs % 3);
  }
  for (int z = zs; z < ze; ++z) {
    if (PIPE) {
      commit((z + 1) % 3);                             // plane z + 1 (loaded during the previous iteration): slot (z - 2) % 3 is free
      __syncthreads();                                 // planes z-1, z, z+1 are in LDS
      if (z + 1 < ze) issue(z + 2);                    // lands while the matrix cores work on plane z
    } else {
      stage(z + 1, (z + 1) % 3);
      __syncthreads();                                 // planes z-1, z, z+1 are in LDS
    }
    // ---- (b) this wave's channels
#pragma unroll
    for (int ch = 0; ch < CPW; ++ch) {
      const int c = wave * CPW + ch;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int q = 2 * s + (kg >> 1);
        bf16x8_t bfrag;
        if (q <= 8) {
          const int dz = q / 3, dy = q % 3;
          const unsigned short* src = ring[(z + dz - 1 + 3) % 3] + c * CH_PITCH + (m + dy) * WIN + (kg & 1) * 8;   // m doubles as n here
          bfrag = *reinterpret_cast<const bf16x8_t*>(src);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) bfrag[i] = (__bf16)0.f;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ch][s], bfrag, acc, 0, 0, 0);
      }
      // D: lane holds rows (x) 4 * kg + r of column (y) n = lane & 15
      const int nrow = lane & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mx = kg * 4 + r;
        if (mx < TX) {
          const unsigned short h = f2bf(acc[r] + bv[ch]);
          ostage[(nrow * TX + mx) * C + c] = h;
          if (y0 + nrow < g.H && x0 + mx < g.W) { const float v = bf2f(h); s1[ch] += v; s2[ch] += v * v; }
        }
      }
    }
    __syncthreads();                                   // the output tile of plane z is complete
    for (int chunk = tid; chunk < TY * TX * (C / 8); chunk += 256) {
      const int part = chunk & 3, vox = chunk >> 2;
      const int row = vox / TX, col = vox % TX;
      if (y0 + row < g.H && x0 + col < g.W)
        *reinterpret_cast<u32x4_t*>(yn + (long)z * plane_elems + ((long)(y0 + row) * g.W + x0 + col) * C + part * 8) =
            *reinterpret_cast<const u32x4_t*>(ostage + vox * C + part * 8);
    }
    __syncthreads();                                   // ostage and ring slot (z - 1) % 3 are free again
  }
  // ---- statistics: per workgroup and channel (sum, sum of squares) -- wave-level reduction through shuffles
#pragma unroll
  for (int ch = 0; ch < CPW; ++ch) {
    float a = s1[ch], q2 = s2[ch];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); q2 += __shfl_xor(q2, off); }
    if (lane == 0) {
      stats[((long)blockIdx.x * 2 + 0) * C + wave * CPW + ch] = a;
      stats[((long)blockIdx.x * 2 + 1) * C + wave * CPW + ch] = q2;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static unsigned short h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float h_bf2f(unsigned short h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

static Geom make_geom(int N, int D, int H, int W) {
  Geom g{N, D, H, W, 0, 0, (H + TY - 1) / TY, (W + TX - 1) / TX};
  int nzc = (1024 + g.ty * g.tx - 1) / (g.ty * g.tx);      // >= ~1024 workgroups per sample, chunks of >= 14 planes
  int maxc = D / 14 < 1 ? 1 : D / 14;
  if (nzc > maxc) nzc = maxc;
  g.zc = (D + nzc - 1) / nzc;
  g.nzc = (D + g.zc - 1) / g.zc;
  return g;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

static double run_case(int N, int D, int H, int W, bool check, int reps, bool pipe) {
  auto kernel = pipe ? toeplitz_dwconv_kernel<true> : toeplitz_dwconv_kernel<false>;
  const long vox = (long)N * D * H * W, elems = vox * C;
  std::vector<unsigned short> hx(elems), hy(elems);
  std::vector<float> hw(27 * C), hb(C);
  uint32_t seed = 12345u + (uint32_t)(D * 31 + W);
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (long i = 0; i < elems; ++i) hx[i] = h_f2bf(2.0f * rnd());
  for (int i = 0; i < 27 * C; ++i) hw[i] = h_bf2f(h_f2bf(0.6f * rnd()));     // taps rounded to bf16 (what the MFMA consumes)
  for (int i = 0; i < C; ++i) hb[i] = rnd();
  unsigned short *dx, *dy; float *dw, *db, *dst;
  Geom g = make_geom(N, D, H, W);
  const long wgs = (long)N * g.nzc * g.ty * g.tx;
  CK(hipMalloc(&dx, elems * 2)); CK(hipMalloc(&dy, elems * 2)); CK(hipMalloc(&dw, 27 * C * 4)); CK(hipMalloc(&db, C * 4));
  CK(hipMalloc(&dst, wgs * 2 * C * 4));
  CK(hipMemcpy(dx, hx.data(), elems * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), 27 * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0, elems * 2));
  hipLaunchKernelGGL(kernel, dim3((unsigned)wgs), dim3(256), 0, 0, dx, dy, dw, db, dst, g);
  CK(hipDeviceSynchronize());
  double ms = 0.0;
  if (reps > 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3((unsigned)wgs), dim3(256), 0, 0, dx, dy, dw, db, dst, g);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); ms = t / reps;
  }
  if (check) {
    CK(hipMemcpy(hy.data(), dy, elems * 2, hipMemcpyDeviceToHost));
    std::vector<float> hst(wgs * 2 * C);
    CK(hipMemcpy(hst.data(), dst, hst.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, sum_ref = 0.0, sum_got = 0.0; long bad = 0;
    for (int nn = 0; nn < N; ++nn) for (int z = 0; z < D; ++z) for (int yy = 0; yy < H; ++yy) for (int xx = 0; xx < W; ++xx) for (int c = 0; c < C; ++c) {
      float acc = hb[c];
      for (int kz = 0; kz < 3; ++kz) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
        const int zz = z + kz - 1, y2 = yy + ky - 1, x2 = xx + kx - 1;
        if (zz < 0 || zz >= D || y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
        acc += hw[((kz * 3 + ky) * 3 + kx) * C + c] * h_bf2f(hx[(((long)(nn * D + zz) * H + y2) * W + x2) * C + c]);
      }
      const float ref = h_bf2f(h_f2bf(acc)), got = h_bf2f(hy[(((long)(nn * D + z) * H + yy) * W + xx) * C + c]);
      const double d = std::fabs((double)ref - got), tol = 0.0079 * std::fabs(ref) + 1e-6;        // one bf16 ulp (summation order)
      if (d > tol) ++bad;
      if (d > worst) worst = d;
      sum_ref += ref;
    }
    for (long i = 0; i < wgs; ++i) for (int c = 0; c < C; ++c) sum_got += hst[(i * 2) * C + c];
    printf("%s check %dx%dx%dx%d: max |d| %.3e, outside one bf16 ulp: %ld of %ld; statistics sum %.4f vs reference %.4f\n", pipe ? "pipelined" : "plain    ", N, D, H, W, worst, bad,
           elems, sum_got, sum_ref);
  }
  CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dst));
  return ms;
}

int main(int argc, char** argv) {
  for (int pipe = 0; pipe < 2; ++pipe) {
    run_case(1, 9, 20, 31, true, 0, pipe != 0);
    run_case(2, 16, 32, 28, true, 0, pipe != 0);
  }
  if (argc > 1 && std::string(argv[1]) == "check") return 0;
  const int batch = argc > 2 ? atoi(argv[2]) : 8;                       // `time 2`: a 2-window batch (0.36 GB of traffic: partly cache-resident)
  const double gb = 2.0 * batch * 112.0 * 112 * 112 * C * 2 / 1e9;
  for (int pipe = 0; pipe < 2; ++pipe) {
    const double ms = run_case(batch, 112, 112, 112, false, 10, pipe != 0);
    printf("%s %d x 112^3 x 32 bf16: %.1f us per launch, %.2f TB/s algorithmic (x + y once); the z-march VALU kernel: ~0.53 ms per 8 windows at level 0\n",
           pipe ? "pipelined" : "plain    ", batch, ms * 1e3, gb / ms);
  }
  return 0;
}
