// Dense Conv3d with a stride and dense ConvTranspose3d (gather form) on MFMA, plus their weight gradient -- the
// resampling convolutions of the MONAI-style residual U-Net (reference models/architectures/monai_models.py:197-250:
// ResidualUnit down layers with stride 2, `Convolution(is_transposed=True)` up layers, kernel 3, padding 1,
// output_padding stride-1).  Same transposed implicit GEMM as conv3d_kernels.hip
//     Y^T[o][v] = sum_{tap,c} W[o][tap][c] * f(X)^T[c][src(v, tap)]
// only the source voxel of (output voxel v, tap) differs:
//     strided conv      src = v * s + tap - pad                      (zero outside the input)
//     transposed conv   src = (v + pad - tap) / s  when divisible    (else the tap contributes nothing)
// so one kernel serves: strided forward, transposed forward, the data gradient of a strided conv (= transposed gather
// with the forward weights read [C_out][C_in] -> [in][out]) and the data gradient of a transposed conv (= strided conv).
// The fused pre-activation f(x) = act(a[n][c] * x + b[n][c]) is available exactly as in the stride-1 kernel.
#include "pw_common.h"

namespace pytc {

// conv3d_kernels.hip: the scalar-field stencil for one input channel (3^3 taps, stride 1 / 2)
bool conv_c1_stencil_try(const void* x, const void* wp, const float* bias, const float* ab, int act_in, const EpiParams& e, int N,
                         int Do, int Ho, int Wo, int Di, int Hi, int Wi, int C_in, int C_out, int kd, int kh, int kw, int stride, int pad,
                         int dtype, hipStream_t s);

struct SConvParams {
  const void* x;
  const void* wp;       // [mtile][tap][kgroup][lane][EPL]  (conv3d_pack_kernel layout)
  const float* bias;
  const float* ab;      // [N][2][C_in] or NULL
  EpiParams e;
  int N, C_in, C_out, KG, MTt;
  int Do, Ho, Wo;       // output grid
  int Di, Hi, Wi;       // input grid
  int kd, kh, kw;
  int sd, sh, sw;       // stride per axis
  int pd, ph, pw;       // padding per axis
  int transposed;
  int phase_major;      // transposed, stride 2 on every axis, output grid = 2 x input grid: rows are enumerated phase by phase (below)
  int act_in;
  float act_param;
};

__device__ __forceinline__ float s_pre_act(float v, int act, float prm) {
  switch (act) {
    case PYTC_ACT_RELU: return fmaxf(v, 0.f);
    case PYTC_ACT_LEAKY: return v > 0.f ? v : v * prm;
    case PYTC_ACT_ELU: return v > 0.f ? v : prm * (__expf(v) - 1.0f);
    default: return v;
  }
}

// source coordinate along one axis; returns false when the tap does not touch the input
__device__ __forceinline__ bool src_coord(int o, int t, int s, int pad, int n_in, int transposed, int& i) {
  if (!transposed) {
    i = o * s + t - pad;
    return i >= 0 && i < n_in;
  }
  const int num = o + pad - t;
  if (num < 0) return false;
  i = num / s;
  return (num - i * s) == 0 && i < n_in;
}

template <typename TI, typename TW, typename TO, int MT, int NT>
__global__ void __launch_bounds__(256)
conv3d_strided_kernel(SConvParams p) {
  typedef Mma<TW> M;
  constexpr int EPL = M::EPL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.z;
  const int mt0 = blockIdx.y * MT;
  const long rps = (long)p.Do * p.Ho * p.Wo;
  const long rps_in = (long)p.Di * p.Hi * p.Wi;
  const long row0 = ((long)blockIdx.x * 4 + wave) * (NT * 16);
  if (row0 >= rps) return;
  const int r = lane & 15, kb = lane >> 4;

  // Round 6: a stride-2 transposed gather enumerates its output rows PHASE BY PHASE -- row = phase * (Di Hi Wi) + input-grid position,
  // output voxel (2 iz + a, 2 iy + b, 2 ix + c) for phase (a, b, c).  The taps an output voxel sees depend only on its phase (per
  // axis: one tap for an even coordinate, two for an odd one -- 27 taps spread over 8 phases), so all 16 rows of a fragment, and all
  // NT fragments of a wave, use the SAME taps: the wave-uniform skip below drops the others (27 / 8 = 3.4 tap iterations per row on
  // average).  In raster order every wave held both x parities: 6.75 iterations of which half the lanes contributed zeros.
  // Same taps in the same ascending order per output voxel -> same bits.
  long orow[NT];
  int vz[NT], vy[NT], vx[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    long o = row0 + nt * 16 + r;
    long oc = o < rps ? o : rps - 1;
    if (p.phase_major) {
      const int phase = (int)(oc / rps_in);
      long q = oc - (long)phase * rps_in;
      const int ix = (int)(q % p.Wi);
      q /= p.Wi;
      vx[nt] = 2 * ix + (phase & 1);
      vy[nt] = 2 * (int)(q % p.Hi) + ((phase >> 1) & 1);
      vz[nt] = 2 * (int)(q / p.Hi) + (phase >> 2);
      orow[nt] = o < rps ? ((long)vz[nt] * p.Ho + vy[nt]) * p.Wo + vx[nt] : rps;
    } else {
      orow[nt] = o;
      vx[nt] = (int)(oc % p.Wo);
      long t = oc / p.Wo;
      vy[nt] = (int)(t % p.Ho);
      vz[nt] = (int)(t / p.Ho);
    }
  }
  const TI* xn = reinterpret_cast<const TI*>(p.x) + (long)n * rps_in * p.C_in;
  const bool vec_ok = (p.C_in % EPL) == 0;
  const bool affine = p.ab != nullptr;

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const typename M::frag_t* wp = reinterpret_cast<const typename M::frag_t*>(p.wp);
  const int ntap = p.kd * p.kh * p.kw;
  for (int kg = 0; kg < p.KG; ++kg) {
    const int k0 = kg * M::KSTEP + kb * EPL;
    float av[EPL], bv[EPL];
    if (affine) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        bool ok = k0 + j < p.C_in;
        av[j] = ok ? p.ab[((long)n * 2 + 0) * p.C_in + k0 + j] : 0.f;
        bv[j] = ok ? p.ab[((long)n * 2 + 1) * p.C_in + k0 + j] : 0.f;
      }
    }
    for (int tap = 0; tap < ntap; ++tap) {
      const int tx = tap % p.kw;
      const int tt = tap / p.kw;
      const int ty = tt % p.kh;
      const int tz = tt / p.kh;
      typename M::frag_t bf[NT];
      bool any = false;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int z, y, x;
        const bool ok = src_coord(vz[nt], tz, p.sd, p.pd, p.Di, p.transposed, z) &
                        src_coord(vy[nt], ty, p.sh, p.ph, p.Hi, p.transposed, y) &
                        src_coord(vx[nt], tx, p.sw, p.pw, p.Wi, p.transposed, x);
        float v[EPL];
        if (ok) {
          load_row_frag<TI, EPL>(xn + (((long)z * p.Hi + y) * p.Wi + x) * p.C_in, k0, p.C_in, vec_ok, v);
          if (affine) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) v[j] = fmaf(v[j], av[j], bv[j]);
          }
          if (p.act_in != PYTC_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) v[j] = (k0 + j < p.C_in) ? s_pre_act(v[j], p.act_in, p.act_param) : 0.f;
          }
          any = true;
        } else {
#pragma unroll
          for (int j = 0; j < EPL; ++j) v[j] = 0.f;
        }
        bf[nt] = M::from_floats(v);
      }
      // a transposed conv touches 1/(sd*sh*sw) of the taps per voxel: skip the MFMAs of a tap no lane of the wave uses
      if (!__any(any)) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (mt0 + mt < p.MTt) {
          typename M::frag_t af = wp[(((long)(mt0 + mt) * ntap + tap) * p.KG + kg) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = M::mma(af, bf[nt], acc[mt][nt]);
        }
      }
    }
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int o0 = (mt0 + mt) * 16 + kb * 4;
    if (o0 >= p.C_out) continue;
    float bo[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) bo[rr] = (p.bias && o0 + rr < p.C_out) ? p.bias[o0 + rr] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (orow[nt] >= rps) continue;
      float v[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) v[rr] = acc[mt][nt][rr] + bo[rr];
      finish_and_store<TO, 4>(v, p.e, n, orow[nt], o0);
    }
  }
}

template <typename TW>
__global__ void __launch_bounds__(256)
conv3d_pack_direct_kernel(const float* __restrict__ w, int C_out, int C_in, int ntap, TW* __restrict__ packed, int KG,
                          long total, long s_o, long s_c, int flip) {
  typedef Mma<TW> M;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int j = (int)(i % M::EPL);
  long t = i / M::EPL;
  int lane = (int)(t % 64); t /= 64;
  int kg = (int)(t % KG); t /= KG;
  int tap = (int)(t % ntap);
  int mt = (int)(t / ntap);
  int o = mt * 16 + (lane & 15);
  int k = kg * M::KSTEP + (lane >> 4) * M::EPL + j;
  float v = 0.f;
  if (o < C_out && k < C_in) v = w[o * s_o + k * s_c + (flip ? ntap - 1 - tap : tap)];
  packed[i] = from_f32<TW>(v);
}

// ---- weight gradient of the strided / transposed conv (VALU, fp32 accumulation, per-slot partials, fixed-order sum) -----
//   dW[o][k][tap] = sum_{rows r of the SMALL grid}  G[r][o] * A[src(r, tap)][k]
// strided conv:    G = dY on the output grid (C_out channels), A = the conv input on the input grid;  src = r*s + tap - pad
// transposed conv: G = the conv INPUT x (low-res grid, C_in_T channels), A = dY (high-res grid, C_out_T channels): the result
//                  [C_in_T][C_out_T][tap] is ConvTranspose3d's weight layout.
struct SwGeom { int Ds, Hs, Ws, Db, Hb, Wb, kd, kh, kw, sd, sh, sw, pd, ph, pw; };
constexpr int SW_TO = 64, SW_TK = 64, SW_TR = 32;
template <typename T>
__global__ void __launch_bounds__(256)
conv3d_wgrad_strided_kernel(const T* __restrict__ a, const T* __restrict__ gsm, float* __restrict__ dWp, long rows_total,
                            SwGeom g, int C_k, int C_o, long rows_per_slot, int slots) {
  __shared__ float sa[SW_TR][SW_TK + 1];
  __shared__ float sd[SW_TR][SW_TO + 1];
  __shared__ long rsrc[SW_TR];                 // element offset of the row's source voxel in `a` (-1: outside / past the slot)
  const int slot = blockIdx.x, tap = blockIdx.z;
  const int tiles_k = (C_k + SW_TK - 1) / SW_TK;
  const int o_base = (blockIdx.y / tiles_k) * SW_TO, k_base = (blockIdx.y % tiles_k) * SW_TK;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int tz_ = tap / (g.kh * g.kw), ty_ = (tap / g.kw) % g.kh, tx_ = tap % g.kw;
  const long vol_s = (long)g.Ds * g.Hs * g.Ws;
  const long vol_b = (long)g.Db * g.Hb * g.Wb;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const long r_begin = (long)slot * rows_per_slot;
  const long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  for (long r0 = r_begin; r0 < r_end; r0 += SW_TR) {
    // the source voxel of a row is the same for all its channels: decode it ONCE per row (the 64-bit divisions used to run
    // per staged element -- 64x per row -- and dominated the kernel)
    if (threadIdx.x < SW_TR) {
      const long r = r0 + threadIdx.x;
      long off = -1;
      if (r < r_end) {
        const long n = r / vol_s, rem = r % vol_s;
        const int x = (int)(rem % g.Ws), y = (int)((rem / g.Ws) % g.Hs), z = (int)(rem / ((long)g.Ws * g.Hs));
        const int bz = z * g.sd + tz_ - g.pd, by = y * g.sh + ty_ - g.ph, bx = x * g.sw + tx_ - g.pw;
        if (bz >= 0 && bz < g.Db && by >= 0 && by < g.Hb && bx >= 0 && bx < g.Wb)
          off = (n * vol_b + ((long)bz * g.Hb + by) * g.Wb + bx) * C_k;
      }
      rsrc[threadIdx.x] = off;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SW_TR * SW_TK; i += 256) {
      const int rr = i / SW_TK, kk = i % SW_TK;
      const long off = rsrc[rr];
      sa[rr][kk] = (off >= 0 && k_base + kk < C_k) ? to_f32<T>(a[off + k_base + kk]) : 0.f;
    }
    for (int i = threadIdx.x; i < SW_TR * SW_TO; i += 256) {
      const int rr = i / SW_TO, oo = i % SW_TO;
      const long r = r0 + rr;
      sd[rr][oo] = (r < r_end && o_base + oo < C_o) ? to_f32<T>(gsm[r * C_o + o_base + oo]) : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < SW_TR; ++rr) {
      float dv[4], xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { dv[i] = sd[rr][ty * 4 + i]; xv[i] = sa[rr][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], xv[j], acc[i][j]);
    }
    __syncthreads();
  }
  const int taps = g.kd * g.kh * g.kw;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o_base + ty * 4 + i;
    if (o >= C_o) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k_base + tx * 4 + j;
      if (k < C_k) dWp[(((long)slot * taps + tap) * C_o + o) * C_k + k] = acc[i][j];
    }
  }
}


// C_k <= 4 (the first conv of a U-Net, 1 -> C; the last transposed conv, C -> 1): the 64 x 64 tile above would spend 63 / 64 of
// its staging and FMAs on padding (3.4 ms per launch for 1 -> 32 at 2 x 24 x 256 x 256).  Here a thread owns one output channel
// o and every 4th row of its slot: G[r][o] is a coalesced 64-lane load, the row's C_k source values are wave-uniform (broadcast
// loads), the row coordinates advance incrementally (no divisions in the loop); the 4 row lanes meet in LDS in a fixed order.
constexpr int SWT_TO = 64, SWT_RL = 4, SWT_KMAX = 4;
template <typename T>
__global__ void __launch_bounds__(256)
conv3d_wgrad_strided_thin_kernel(const T* __restrict__ a, const T* __restrict__ gsm, float* __restrict__ dWp, long rows_total,
                                 SwGeom g, int C_k, int C_o, long rows_per_slot, int slots) {
  __shared__ float red[SWT_RL][SWT_TO][SWT_KMAX];
  const int slot = blockIdx.x, tap = blockIdx.z;
  const int o = blockIdx.y * SWT_TO + (threadIdx.x % SWT_TO), rl = threadIdx.x / SWT_TO;
  const int tz_ = tap / (g.kh * g.kw), ty_ = (tap / g.kw) % g.kh, tx_ = tap % g.kw;
  const long vol_s = (long)g.Ds * g.Hs * g.Ws;
  const long vol_b = (long)g.Db * g.Hb * g.Wb;
  const long r_begin = (long)slot * rows_per_slot;
  const long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  float acc[SWT_KMAX] = {0.f, 0.f, 0.f, 0.f};
  long r = r_begin + rl;
  if (r < r_end) {
    long n = r / vol_s;
    const long rem = r % vol_s;
    int x = (int)(rem % g.Ws), y = (int)((rem / g.Ws) % g.Hs), z = (int)(rem / ((long)g.Ws * g.Hs));
    const bool o_ok = o < C_o;
    for (; r < r_end; r += SWT_RL) {
      const int bz = z * g.sd + tz_ - g.pd, by = y * g.sh + ty_ - g.ph, bx = x * g.sw + tx_ - g.pw;
      if (bz >= 0 && bz < g.Db && by >= 0 && by < g.Hb && bx >= 0 && bx < g.Wb && o_ok) {     // wave-uniform except o_ok
        const float gv = to_f32<T>(gsm[r * C_o + o]);
        const T* src = a + (n * vol_b + ((long)bz * g.Hb + by) * g.Wb + bx) * C_k;
#pragma unroll
        for (int k = 0; k < SWT_KMAX; ++k)
          if (k < C_k) acc[k] = fmaf(gv, to_f32<T>(src[k]), acc[k]);
      }
      x += SWT_RL;
      while (x >= g.Ws) { x -= g.Ws; if (++y == g.Hs) { y = 0; if (++z == g.Ds) { z = 0; ++n; } } }
    }
  }
#pragma unroll
  for (int k = 0; k < SWT_KMAX; ++k) red[rl][threadIdx.x % SWT_TO][k] = acc[k];
  __syncthreads();
  if (rl == 0 && o < C_o) {
    const int taps = g.kd * g.kh * g.kw;
    for (int k = 0; k < C_k; ++k) {
      const int oo = threadIdx.x % SWT_TO;
      dWp[(((long)slot * taps + tap) * C_o + o) * C_k + k] = ((red[0][oo][k] + red[1][oo][k]) + red[2][oo][k]) + red[3][oo][k];
    }
  }
}

// C_k == 1 and 3 x 3 x 3 taps (the two layers above in a 1-channel network): all 27 taps in one pass -- G[r][o] is read ONCE
// instead of once per tap (27 launches-worth of re-reads), the 27 source values of a row are wave-uniform broadcast loads.
template <typename T>
__global__ void __launch_bounds__(256)
conv3d_wgrad_strided_thin1_kernel(const T* __restrict__ a, const T* __restrict__ gsm, float* __restrict__ dWp, long rows_total,
                                  SwGeom g, int C_o, long rows_per_slot, int slots) {
  __shared__ float red[SWT_RL][SWT_TO][27 + 1];
  const int slot = blockIdx.x;
  const int oo = threadIdx.x % SWT_TO, o = blockIdx.y * SWT_TO + oo, rl = threadIdx.x / SWT_TO;
  const long vol_s = (long)g.Ds * g.Hs * g.Ws;
  const long vol_b = (long)g.Db * g.Hb * g.Wb;
  const long r_begin = (long)slot * rows_per_slot;
  const long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  long r = r_begin + rl;
  if (r < r_end) {
    long n = r / vol_s;
    const long rem = r % vol_s;
    int x = (int)(rem % g.Ws), y = (int)((rem / g.Ws) % g.Hs), z = (int)(rem / ((long)g.Ws * g.Hs));
    const bool o_ok = o < C_o;
    for (; r < r_end; r += SWT_RL) {
      const float gv = o_ok ? to_f32<T>(gsm[r * C_o + o]) : 0.f;
      const T* base = a + n * vol_b;
      const int bz0 = z * g.sd - g.pd, by0 = y * g.sh - g.ph, bx0 = x * g.sw - g.pw;
#pragma unroll
      for (int tz = 0; tz < 3; ++tz) {
        const int bz = bz0 + tz;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int by = by0 + ty;
          const bool row_ok = bz >= 0 && bz < g.Db && by >= 0 && by < g.Hb;
          const T* rowp = base + ((long)min(max(bz, 0), g.Db - 1) * g.Hb + min(max(by, 0), g.Hb - 1)) * g.Wb;
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const int bx = bx0 + tx;
            const float av = (row_ok && bx >= 0 && bx < g.Wb) ? to_f32<T>(rowp[bx]) : 0.f;      // wave-uniform address
            acc[(tz * 3 + ty) * 3 + tx] = fmaf(gv, av, acc[(tz * 3 + ty) * 3 + tx]);
          }
        }
      }
      x += SWT_RL;
      while (x >= g.Ws) { x -= g.Ws; if (++y == g.Hs) { y = 0; if (++z == g.Ds) { z = 0; ++n; } } }
    }
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) red[rl][oo][t] = acc[t];
  __syncthreads();
  if (rl == 0 && o < C_o) {
    for (int t = 0; t < 27; ++t)
      dWp[((long)slot * 27 + t) * C_o + o] = ((red[0][oo][t] + red[1][oo][t]) + red[2][oo][t]) + red[3][oo][t];
  }
}

// C_k == 1, 3 x 3 x 3 taps, stride 2, padding 1 as a REDUCTION over staged lines (round 5).  The kernel above walks its rows with 27
// dependent broadcast loads each (499 us for 2 x 12 x 128 x 128 x 32 gradients: 30 MB of operands, 0.0005 of anything).  Here a
// workgroup takes whole x-lines of the small grid: the 9 source lines (3 z x 3 y of the 1-channel input, zero padded, as fp32) and the
// gradient line ([Ws][C_o] as fp32) are staged in LDS with coalesced loads, then thread (tap slot t / 32, channel t % 32) runs along
// the line: one LDS read of its gradient value(s), one broadcast LDS read and one FMA per tap it owns (taps slot, slot + 8, slot + 16,
// slot + 24).  One partial [27][C_o] per workgroup (slot order fixed: deterministic), sw_reduce_slots_kernel finishes.
template <typename T, int COG>
__global__ void __launch_bounds__(256)
conv3d_wgrad_s2_c1_line_kernel(const T* __restrict__ a, const T* __restrict__ gsm, float* __restrict__ dWp, int N, SwGeom g, int C_o,
                               int lines_per_slot) {
  extern __shared__ float lds_f[];
  const int WP = g.Wb + 2;                                   // column bx lives at index bx + 1 (bx = -1 .. Wb)
  float* bl = lds_f;                                         // [9][WP]
  float* gl = lds_f + 9 * WP;                                // [Ws][C_o]
  const int tid = threadIdx.x, o = tid & 31, ts = tid >> 5;  // channel (+ 32 c), tap slot 0..7
  const long lines_total = (long)N * g.Ds * g.Hs;
  const long l_begin = (long)blockIdx.x * lines_per_slot;
  const long l_end = l_begin + lines_per_slot < lines_total ? l_begin + lines_per_slot : lines_total;
  float acc[4][COG];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < COG; ++c) acc[k][c] = 0.f;
  int lrow[4], lcol[4];                                      // this thread's taps: source line (tz * 3 + ty) and column offset tx
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int tap = ts + 8 * k; lrow[k] = tap < 27 ? tap / 3 : 0; lcol[k] = tap % 3; }
  for (long line = l_begin; line < l_end; ++line) {
    const int y = (int)(line % g.Hs);
    const int z = (int)((line / g.Hs) % g.Ds);
    const long n = line / ((long)g.Hs * g.Ds);
    __syncthreads();                                          // the previous line's reads are done
    for (int i = tid; i < 9 * WP; i += 256) {
      const int l = i / WP, bx = i % WP - 1;
      const int bz = 2 * z - 1 + l / 3, by = 2 * y - 1 + l % 3;
      const bool ok = bz >= 0 && bz < g.Db && by >= 0 && by < g.Hb && bx >= 0 && bx < g.Wb;
      bl[i] = ok ? to_f32<T>(a[((n * g.Db + bz) * g.Hb + by) * g.Wb + bx]) : 0.f;
    }
    const T* gline = gsm + line * g.Ws * C_o;
    for (int i = tid; i < g.Ws * C_o; i += 256) gl[i] = to_f32<T>(gline[i]);
    __syncthreads();
    for (int x = 0; x < g.Ws; ++x) {
      float gv[COG];
#pragma unroll
      for (int c = 0; c < COG; ++c) gv[c] = gl[x * C_o + o + 32 * c];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float av = bl[lrow[k] * WP + 2 * x + lcol[k]];           // source column 2x + tx - 1, stored at + 1
#pragma unroll
        for (int c = 0; c < COG; ++c) acc[k][c] = fmaf(gv[c], av, acc[k][c]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int tap = ts + 8 * k;
    if (tap < 27) {
#pragma unroll
      for (int c = 0; c < COG; ++c) dWp[((long)blockIdx.x * 27 + tap) * C_o + o + 32 * c] = acc[k][c];
    }
  }
}

// out[i] = sum_s part[s][i], slots in order within 16 interleaved lanes, then the lanes in order (fixed tree)
__global__ void __launch_bounds__(256)
sw_reduce_slots_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int slots) {
  __shared__ float sm[16][17];
  const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + e;
  float a = 0.f;
  if (i < n)
    for (int s = j; s < slots; s += 16) a += part[(long)s * n + i];
  sm[j][e] = a;
  __syncthreads();
  if (j == 0 && i < n) {
    float t = sm[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += sm[q][e];
    out[i] = t;
  }
}

template <typename TI, typename TW, typename TO>
static void launch_sconv(const SConvParams& p, hipStream_t s) {
  constexpr int NT = 4;
  const long rps = (long)p.Do * p.Ho * p.Wo;
  const int MT = p.MTt >= 4 ? 4 : (p.MTt >= 2 ? 2 : 1);
  dim3 grid((unsigned)((rps + 4L * NT * 16 - 1) / (4L * NT * 16)), (unsigned)((p.MTt + MT - 1) / MT), (unsigned)p.N);
  dim3 block(256);
  switch (MT) {
    case 1: hipLaunchKernelGGL((conv3d_strided_kernel<TI, TW, TO, 1, NT>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((conv3d_strided_kernel<TI, TW, TO, 2, NT>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((conv3d_strided_kernel<TI, TW, TO, 4, NT>), grid, block, 0, s, p); break;
  }
}

// ---- ConvTranspose3d(k 3, stride 2, padding 1, output_padding 1) with C_out <= 4: the last up-sampling layer of a U-Net whose
// output has one (or a few) channels.  On the MFMA kernel above its single output channel is a 16-row tile and every one of
// the 27 taps is decoded per voxel (1.9 ms for 64 -> 1 at 2 x 24 x 256 x 256).  Here a thread owns one output voxel: along each
// axis an even coordinate has ONE contributing tap (t = 1, source o / 2) and an odd one two (t = 0 / 2, sources (o + 1) / 2 and
// (o - 1) / 2), so 1 .. 8 of the 27 taps are visited; the source row is read in 16-byte channel chunks, the taps come from LDS
// ([tap][o][c] fp32).  w: ConvTranspose3d layout [C_in][C_out][27] fp32.
// ONE output channel (round 6): the output-voxel form below reads the full C_in-channel row of 1 .. 8 source voxels per output voxel
// (3.4 on average, every source row fetched by up to 27 outputs: 304 us for 64 -> 1 at 2 x 24 x 256 x 256; the eight-phase tile kernel:
// 180 us, bound by its staging).  Input-centric instead: (1) every INPUT voxel forms its 27 tap products once,
// P[t][i] = sum_c w[c][t] * x[i][c] (planar fp32, 27 x rows), (2) every output voxel sums the 1 .. 8 entries its parity selects.
template <typename T>
__global__ void __launch_bounds__(256)
convT3d_c1_dots_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ P, long rows, int C_in) {
  extern __shared__ float wl[];                            // [27][C_in]
  for (int i = threadIdx.x; i < 27 * C_in; i += 256) wl[i] = w[(long)(i % C_in) * 27 + i / C_in];
  __syncthreads();
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  const T* row = x + r * C_in;
  for (int c = 0; c < C_in; c += 8) {
    float xv[8];
    VecIO<T, 8>::load(row + c, xv);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wl + t * C_in + c);
      const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(wl + t * C_in + c + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[t] = fmaf(xv[j], w0[j], acc[t]); acc[t] = fmaf(xv[4 + j], w1[j], acc[t]); }
    }
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) P[(long)t * rows + r] = acc[t];
}

template <typename T>
__global__ void __launch_bounds__(256)
convT3d_c1_gather_kernel(const float* __restrict__ P, const float* __restrict__ bias, T* __restrict__ y, int N, int Di, int Hi, int Wi) {
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const long total = (long)N * Do * Ho * Wo, rows = (long)N * Di * Hi * Wi;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= total) return;
  const int ox = (int)(v % Wo);
  long t = v / Wo;
  const int oy = (int)(t % Ho); t /= Ho;
  const int oz = (int)(t % Do);
  const long n = t / Do;
  // per axis: even o -> (tap 1, source o / 2); odd o -> (tap 2, (o - 1) / 2) and (tap 0, (o + 1) / 2) when that source exists;
  // accumulation in ascending tap order (the order of the gather form)
  int tz[2], sz[2], ty[2], sy[2], tx[2], sx[2];
  auto axis = [](int o, int n_in, int (&tp)[2], int (&sp)[2]) {
    if ((o & 1) == 0) { tp[0] = 1; sp[0] = o >> 1; tp[1] = -1; sp[1] = 0; }
    else {
      tp[0] = 0; sp[0] = (o + 1) >> 1; tp[1] = 2; sp[1] = (o - 1) >> 1;
      if (sp[0] >= n_in) tp[0] = -1;
    }
  };
  axis(oz, Di, tz, sz); axis(oy, Hi, ty, sy); axis(ox, Wi, tx, sx);
  float acc = bias ? bias[0] : 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (tz[a] < 0 || ty[b] < 0 || tx[c] < 0) continue;
        const int tap = (tz[a] * 3 + ty[b]) * 3 + tx[c];
        acc += P[(long)tap * rows + ((n * Di + sz[a]) * Hi + sy[b]) * Wi + sx[c]];
      }
  y[v] = from_f32<T>(acc);
}

constexpr int CT_OMAX = 4;
template <typename T>
__global__ void __launch_bounds__(256)
convT3d_thin_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y,
                    int N, int Di, int Hi, int Wi, int C_in, int C_out) {
  extern __shared__ float wl[];                            // [27][C_out][C_in]
  for (int i = threadIdx.x; i < 27 * C_out * C_in; i += 256) {
    const int c = i % C_in, o = (i / C_in) % C_out, t = i / (C_in * C_out);
    wl[i] = w[((long)c * C_out + o) * 27 + t];
  }
  __syncthreads();
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const long total = (long)N * Do * Ho * Wo;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
    const int ox = (int)(v % Wo);
    long t = v / Wo;
    const int oy = (int)(t % Ho); t /= Ho;
    const int oz = (int)(t % Do);
    const long n = t / Do;
    float acc[CT_OMAX];
#pragma unroll
    for (int o = 0; o < CT_OMAX; ++o) acc[o] = (bias && o < C_out) ? bias[o] : 0.f;
    // candidate taps per axis: (tap, source) pairs; -1 marks "none"
    int tz[2], sz[2], ty[2], sy[2], tx[2], sx[2];
    auto axis = [](int o, int n_in, int (&tp)[2], int (&sp)[2]) {
      if ((o & 1) == 0) { tp[0] = 1; sp[0] = o >> 1; tp[1] = -1; sp[1] = 0; }
      else {
        tp[0] = 0; sp[0] = (o + 1) >> 1; tp[1] = 2; sp[1] = (o - 1) >> 1;
        if (sp[0] >= n_in) tp[0] = -1;
      }
    };
    axis(oz, Di, tz, sz); axis(oy, Hi, ty, sy); axis(ox, Wi, tx, sx);
    for (int a = 0; a < 2; ++a) {
      if (tz[a] < 0) continue;
      for (int b = 0; b < 2; ++b) {
        if (ty[b] < 0) continue;
        for (int c2 = 0; c2 < 2; ++c2) {
          if (tx[c2] < 0) continue;
          const int tap = (tz[a] * 3 + ty[b]) * 3 + tx[c2];
          const T* row = x + (((n * Di + sz[a]) * Hi + sy[b]) * Wi + sx[c2]) * C_in;
          const float* wt = wl + (long)tap * C_out * C_in;
          for (int c = 0; c < C_in; c += 8) {
            float xv[8];
            VecIO<T, 8>::load(row + c, xv);
#pragma unroll
            for (int o = 0; o < CT_OMAX; ++o) {
              if (o < C_out) {
                const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wt + o * C_in + c);
                const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(wt + o * C_in + c + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[o] = fmaf(xv[j], w0[j], acc[o]); acc[o] = fmaf(xv[4 + j], w1[j], acc[o]); }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < CT_OMAX; ++o)
      if (o < C_out) y[v * C_out + o] = from_f32<T>(acc[o]);
  }
}


// ---- stride-2 weight gradient on the matrix cores (bf16, k = 3^3, stride 2, pad 1 on every axis) ------------------------------
// dW[tap][o][k] = sum_r small[r][o] * big[2r + tap - 1][k]  (strided conv: small = dY, big = the conv's input; transposed conv:
// small = the conv's input, big = dY).  Same scheme as conv3d_wgrad_mfma_kernel (csrc/rsunet_train_kernels.hip): the reduction
// runs over voxels, both operands are [row][channel] in HBM while an MFMA fragment wants 8 reduction indices of one channel per
// lane -> 16-byte global loads into a wave-private, row-padded LDS image, gfx950's ds_read_b64_tr_b16 transpose read, then
// v_mfma_f32_16x16x32_bf16.  A wave's unit is one 32-voxel x segment of a line of the SMALL grid, for ONE (dz, dy) of the stencil:
// the big line (z, y) = (2 zs + dz - 1, 2 ys + dy - 1) is staged DE-INTERLEAVED -- odd image O[j] = big[2 (x0 + j) - 1] (33 rows),
// even image E[i] = big[2 (x0 + i)] (32 rows) -- so the three x taps read 32 CONSECUTIVE LDS rows each: dx = 0 -> O[0..31],
// dx = 1 -> E[0..31], dx = 2 -> O[1..32].  One staged line, three taps, 3 x MT x NT MFMAs per unit; the next unit's loads are
// in flight during them; no workgroup barrier in the loop.  Workgroup = (row slot, 32 x 32 (o, k) tile, (dz, dy)); the blocks of
// a slot go to one XCD back to back so the 9 x tiles re-reads of the slot's rows hit that XCD's L2.  Per-slot partials, fixed-order
// cross-wave and cross-slot sums: deterministic.  Replaces the VALU kernel above for the wide layers (64 x 128: 2.3 ms).
template <int MT, int NT>
__global__ void __launch_bounds__(256, 3)
conv3d_wgrad_s2_mfma_kernel(const bf16_t* __restrict__ big, const bf16_t* __restrict__ small, float* __restrict__ dWp, int N,
                            SwGeom g, int C_k, int C_o, long units_per_slot, int slots, int tiles) {
  constexpr int BM = MT * 16, BN = NT * 16;
  constexpr int SG = BM * 2 + 32, SX = BN * 2 + 32;        // LDS row pitch in bytes
  constexpr int AR = 65, ODD = 0, EVEN = 33;                // staged big rows: O[0..32] then E[0..31]
  constexpr int WAVE_BYTES = 32 * SG + AR * SX;
  constexpr int RED_BYTES = 3 * BM * BN * 4;
  constexpr int LDS_BYTES = 4 * WAVE_BYTES > RED_BYTES ? 4 * WAVE_BYTES : RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  typedef unsigned int q4_t __attribute__((ext_vector_type(4)));      // (arrays of the uint4 STRUCT end up in scratch memory)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* lg = lds + wave * WAVE_BYTES;
  unsigned char* la = lg + 32 * SG;
  const int T = tiles * 9;
  const long b = blockIdx.x;
  const int slot = (int)((b / 8 / T) * 8 + (b % 8));
  if (slot >= slots) return;
  const int t = (int)((b / 8) % T);
  const int zy = t % 9, tile = t / 9;
  const int dzi = zy / 3, dyi = zy % 3;
  const int tiles_k = C_k / BN;
  const int o_base = (tile / tiles_k) * BM, k_base = (tile % tiles_k) * BN;
  const int nseg = (g.Ws + 31) / 32;
  const long units = (long)N * g.Ds * g.Hs * nseg;
  const long u_begin = (long)slot * units_per_slot;
  const long u_end = u_begin + units_per_slot < units ? u_begin + units_per_slot : units;

  constexpr int CHG = BM / 8, RG = 64 / CHG, ITG = 32 / RG;   // small rows: 16-B chunks per row, rows per load, loads
  constexpr int CHX = BN / 8, ITA = (AR * CHX + 63) / 64;     // big line: chunk loads per lane
  const int g_row = lane / CHG, g_chunk = lane % CHG;
  q4_t rg[ITG], ra[ITA];
  const q4_t zero4 = {0u, 0u, 0u, 0u};

  auto fetch = [&](long u) {
    const int xs = (int)(u % nseg);
    const long line = u / nseg;                     // (n * Ds + zs) * Hs + ys
    const int ys = (int)(line % g.Hs);
    const long nz = line / g.Hs;
    const int zs = (int)(nz % g.Ds), n = (int)(nz / g.Ds);
    const int x0 = xs * 32;
    const bf16_t* sl = small + line * g.Ws * (long)C_o + o_base + g_chunk * 8;
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int x = x0 + it * RG + g_row;
      rg[it] = x < g.Ws ? *reinterpret_cast<const q4_t*>(sl + (long)x * C_o) : zero4;
    }
    const int zb = 2 * zs + dzi - 1, yb = 2 * ys + dyi - 1;
    const bool ok = zb >= 0 && zb < g.Db && yb >= 0 && yb < g.Hb;           // wave-uniform
    const bf16_t* bl = big + (((long)n * g.Db + (ok ? zb : 0)) * g.Hb + (ok ? yb : 0)) * g.Wb * (long)C_k + k_base;
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
      const int c = it * 64 + lane;
      const int row = c / CHX, chunk = c % CHX;
      const int xb = row < EVEN ? 2 * (x0 + row) - 1 : 2 * (x0 + row - EVEN);
      ra[it] = (ok && row < AR && xb >= 0 && xb < g.Wb) ? *reinterpret_cast<const q4_t*>(bl + (long)xb * C_k + chunk * 8) : zero4;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < ITG; ++it) *reinterpret_cast<q4_t*>(lg + (it * RG + g_row) * SG + g_chunk * 16) = rg[it];
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
      const int c = it * 64 + lane;
      const int row = c / CHX, chunk = c % CHX;
      if (row < AR) *reinterpret_cast<q4_t*>(la + row * SX + chunk * 16) = ra[it];
    }
  };
  const int fr_row = (lane >> 4) * 4 + ((lane & 15) >> 2), fr_col = (lane & 3) * 8;
  auto frag = [&](unsigned char* base, int pitch, int row0, int tl) -> bf16x8_t {
    unsigned char* p = base + (row0 + fr_row) * pitch + tl * 32 + fr_col;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 16 * pitch));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, both);
  };

  f32x4_t acc[3][MT][NT];
#pragma unroll
  for (int tp = 0; tp < 3; ++tp)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[tp][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const long quarter = (u_end - u_begin + 3) / 4;
  const long w_begin = u_begin + wave * quarter;
  const long w_end = w_begin + quarter < u_end ? w_begin + quarter : u_end;
  long u = w_begin;
  if (u < w_end) fetch(u);
  for (; u < w_end; ++u) {
    stage();
    if (u + 1 < w_end) fetch(u + 1);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    bf16x8_t fa[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) fa[m] = frag(lg, SG, 0, m);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int row0 = dx == 0 ? ODD : (dx == 1 ? EVEN : ODD + 1);
      bf16x8_t fb[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) fb[n] = frag(la, SX, row0, n);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[dx][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[n], acc[dx][m][n], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }

  float* red = reinterpret_cast<float*>(lds);
  const int nn = lane & 15, mg = (lane >> 4) * 4;
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(tp * BM + m * 16 + mg + i) * BN + n * 16 + nn] = acc[tp][m][n][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[tp][m][n][i] += red[(tp * BM + m * 16 + mg + i) * BN + n * 16 + nn];
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int o = o_base + m * 16 + mg + i;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int k = k_base + n * 16 + nn;
            dWp[(((long)slot * 27 + zy * 3 + tp) * C_o + o) * C_k + k] = acc[tp][m][n][i];
          }
        }
  }
}

struct SwMfmaPlan { bool ok; int mt, nt, tiles, slots; long per_slot; };
static SwMfmaPlan sw_mfma_plan(int N, const int32_t* small_dims, int C_k, int C_o, const int32_t* kernel, const int32_t* stride,
                               const int32_t* pad, int dtype) {
  SwMfmaPlan p{};
  p.ok = dtype == PYTC_BF16 && C_k % 16 == 0 && C_o % 16 == 0 && tuning_get("conv_wgrad_s2_mfma", 1) != 0;
  for (int a = 0; a < 3 && p.ok; ++a) p.ok = kernel[a] == 3 && (!stride || stride[a] == 2) && (!pad || pad[a] == 1);
  if (!p.ok) return p;
  p.mt = C_o % 32 == 0 ? 2 : 1;
  p.nt = C_k % 32 == 0 ? 2 : 1;
  p.tiles = (C_o / (p.mt * 16)) * (C_k / (p.nt * 16));
  const long units = (long)N * small_dims[0] * small_dims[1] * ((small_dims[2] + 31) / 32);
  long s = 4096 / ((long)p.tiles * 9);                 // ~4096 workgroups over the launch
  if (s > units / 8) s = units / 8;
  s = (s / 8) * 8;
  p.slots = (int)(s < 8 ? 8 : (s > 256 ? 256 : s));
  p.per_slot = (units + p.slots - 1) / p.slots;
  return p;
}

static int sw_slots(long rows_total, int C_k) {
  if (C_k <= SWT_KMAX) {               // thin kernels: one workgroup per (slot, 64 channels [, tap]) -- many small slots fill the chip
    const long s = rows_total / 512;
    return (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
  }
  const long s = rows_total / 4096;
  return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

}  // namespace pytc

using namespace pytc;

// the line-reduction kernel: one input channel, 32 or 64 gradient channels, k 3 / stride 2 / pad 1 on every axis, lines that fit 64 KB of LDS
static bool sw_c1_line_ok(const SwGeom& g, int C_k, int C_o, int dtype) {
  if (C_k != 1 || (C_o != 32 && C_o != 64) || (dtype != PYTC_BF16 && dtype != PYTC_F32) || tuning_get("conv3d_wgrad_c1_line", 1) == 0) return false;
  if (g.kd != 3 || g.kh != 3 || g.kw != 3 || g.sd != 2 || g.sh != 2 || g.sw != 2 || g.pd != 1 || g.ph != 1 || g.pw != 1) return false;
  return (size_t)(9 * (g.Wb + 2) + g.Ws * C_o) * sizeof(float) <= 64 * 1024 && 2 * (g.Ws - 1) + 2 - 1 < g.Wb + 1;
}

static int s_kstep(int dtype) { return dtype == PYTC_BF16 ? 32 : 16; }

extern "C" int64_t pytc_conv3d_direct_packed_elems(int C_out, int C_in, int kd, int kh, int kw, int dtype) {
  if (C_out < 1 || C_in < 1 || kd < 1 || kh < 1 || kw < 1 || (dtype != PYTC_F32 && dtype != PYTC_BF16)) return -1;
  const int ks = s_kstep(dtype);
  return (int64_t)((C_out + 15) / 16) * 16 * kd * kh * kw * ((C_in + ks - 1) / ks) * ks;
}

extern "C" int pytc_conv3d_pack_weight_direct(const float* w, int C_out, int C_in, int kd, int kh, int kw, int64_t s_o,
                                              int64_t s_c, int flip, void* packed, int dtype, void* stream) {
  PYTC_REQUIRE(w && packed, "conv3d_pack_weight_direct: null pointer");
  const long total = pytc_conv3d_direct_packed_elems(C_out, C_in, kd, kh, kw, dtype);
  PYTC_REQUIRE(total > 0, "conv3d_pack_weight_direct: bad arguments");
  const int KG = (C_in + s_kstep(dtype) - 1) / s_kstep(dtype);
  dim3 grid(ceil_div(total, 256)), block(256);
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(conv3d_pack_direct_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, kd * kh * kw,
                       (bf16_t*)packed, KG, total, (long)s_o, (long)s_c, flip);
  else
    hipLaunchKernelGGL(conv3d_pack_direct_kernel<float>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, kd * kh * kw,
                       (float*)packed, KG, total, (long)s_o, (long)s_c, flip);
  PYTC_LAUNCH_CHECK("conv3d_pack_weight_direct");
  return PYTC_OK;
}

extern "C" int pytc_conv3d_strided_fwd(const pytc_conv3d_args* a, const int32_t* in_dims, const int32_t* stride,
                                       const int32_t* pad, int transposed, void* stream) {
  PYTC_REQUIRE(a && a->x && a->w_packed && a->y && in_dims && stride && pad, "conv3d_strided: null pointer");
  PYTC_REQUIRE(a->N >= 1 && a->D >= 1 && a->H >= 1 && a->W >= 1 && a->C_in >= 1 && a->C_out >= 1, "conv3d_strided: bad shape");
  PYTC_REQUIRE(in_dims[0] >= 1 && in_dims[1] >= 1 && in_dims[2] >= 1 && stride[0] >= 1 && stride[1] >= 1 && stride[2] >= 1 &&
               pad[0] >= 0 && pad[1] >= 0 && pad[2] >= 0, "conv3d_strided: bad geometry");
  PYTC_REQUIRE(a->dtype == PYTC_F32 || a->dtype == PYTC_BF16, "conv3d_strided: bad dtype");
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || (a->res_mode == PYTC_RES_ADD && a->res), "conv3d_strided: bad residual");
  SConvParams p;
  p.x = a->x; p.wp = a->w_packed; p.bias = a->bias; p.ab = a->ab;
  p.N = a->N; p.Do = a->D; p.Ho = a->H; p.Wo = a->W; p.C_in = a->C_in; p.C_out = a->C_out;
  p.Di = in_dims[0]; p.Hi = in_dims[1]; p.Wi = in_dims[2];
  p.KG = (a->C_in + s_kstep(a->dtype) - 1) / s_kstep(a->dtype);
  p.MTt = (a->C_out + 15) / 16;
  p.kd = a->kd; p.kh = a->kh; p.kw = a->kw;
  p.sd = stride[0]; p.sh = stride[1]; p.sw = stride[2];
  p.pd = pad[0]; p.ph = pad[1]; p.pw = pad[2];
  p.transposed = transposed ? 1 : 0;
  p.phase_major = (p.transposed && p.sd == 2 && p.sh == 2 && p.sw == 2 && p.Do == 2 * p.Di && p.Ho == 2 * p.Hi && p.Wo == 2 * p.Wi &&
                   tuning_get("convT_phase_major", 1) != 0) ? 1 : 0;
  p.act_in = a->act_in; p.act_param = a->act_param;
  p.e.res = a->res; p.e.res_low = nullptr; p.e.res_bias = nullptr; p.e.y = a->y;
  p.e.rps_out = (long)a->D * a->H * a->W; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode; p.e.nt = 0;
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!transposed && stride[0] == stride[1] && stride[1] == stride[2] && pad[0] == pad[1] && pad[1] == pad[2] &&
      conv_c1_stencil_try(a->x, a->w_packed, a->bias, a->ab, a->act_in, p.e, a->N, a->D, a->H, a->W, p.Di, p.Hi, p.Wi, a->C_in, a->C_out,
                          a->kd, a->kh, a->kw, stride[0], pad[0], a->dtype, s)) {}
  else if (a->dtype == PYTC_F32) launch_sconv<float, float, float>(p, s);
  else launch_sconv<bf16_t, bf16_t, bf16_t>(p, s);
  PYTC_LAUNCH_CHECK("conv3d_strided");
  return PYTC_OK;
}

extern "C" int64_t pytc_conv3d_wgrad_strided_ws_elems(int N, const int32_t* small_dims, int C_k, int C_o,
                                                      const int32_t* kernel) {
  if (!small_dims || !kernel || N < 1) return -1;
  const long rows_total = (long)N * small_dims[0] * small_dims[1] * small_dims[2];
  int slots = sw_slots(rows_total, C_k);
  // the matrix-core path (bf16, k 3, stride 2, pad 1) has its own slot count; the query does not know dtype / stride: size for both
  const SwMfmaPlan m = sw_mfma_plan(N, small_dims, C_k, C_o, kernel, nullptr, nullptr, PYTC_BF16);
  if (m.ok && m.slots > slots) slots = m.slots;
  return (int64_t)slots * kernel[0] * kernel[1] * kernel[2] * C_o * C_k;
}

extern "C" int pytc_conv3d_wgrad_strided(const void* big, const void* small, float* dW, float* workspace, int N,
                                         const int32_t* big_dims, const int32_t* small_dims, int C_k, int C_o,
                                         const int32_t* kernel, const int32_t* stride, const int32_t* pad, int dtype,
                                         void* stream) {
  PYTC_REQUIRE(big && small && dW && workspace && big_dims && small_dims && kernel && stride && pad && N >= 1,
               "conv3d_wgrad_strided: bad arguments");
  SwGeom g{small_dims[0], small_dims[1], small_dims[2], big_dims[0], big_dims[1], big_dims[2], kernel[0], kernel[1], kernel[2],
           stride[0], stride[1], stride[2], pad[0], pad[1], pad[2]};
  const long rows_total = (long)N * g.Ds * g.Hs * g.Ws;
  int slots = sw_slots(rows_total, C_k);
  const long rps = (rows_total + slots - 1) / slots;
  const int taps = g.kd * g.kh * g.kw;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(slots, ((C_o + SW_TO - 1) / SW_TO) * ((C_k + SW_TK - 1) / SW_TK), taps), block(256);
  const SwMfmaPlan mp = sw_mfma_plan(N, small_dims, C_k, C_o, kernel, stride, pad, dtype);
  if (mp.ok && C_k > SWT_KMAX && big_dims[0] >= 1) {
    slots = mp.slots;
    const long blocks = (long)((slots + 7) / 8) * 8 * mp.tiles * 9;
#define SW_MFMA(MT, NT) hipLaunchKernelGGL((conv3d_wgrad_s2_mfma_kernel<MT, NT>), dim3((unsigned)blocks), dim3(256), 0, s, \
                                           (const bf16_t*)big, (const bf16_t*)small, workspace, N, g, C_k, C_o, mp.per_slot, slots, mp.tiles)
    if (mp.mt == 2 && mp.nt == 2) SW_MFMA(2, 2);
    else if (mp.mt == 2) SW_MFMA(2, 1);
    else if (mp.nt == 2) SW_MFMA(1, 2);
    else SW_MFMA(1, 1);
#undef SW_MFMA
  } else if (sw_c1_line_ok(g, C_k, C_o, dtype)) {
    // slot = a run of whole x-lines; the workspace query (pytc_conv3d_wgrad_strided_ws_elems) sizes for sw_slots(): never more here
    const long lines_total = (long)N * g.Ds * g.Hs;
    int lslots = (int)(lines_total < slots ? lines_total : slots);
    const int lps = (int)((lines_total + lslots - 1) / lslots);
    lslots = (int)((lines_total + lps - 1) / lps);
    slots = lslots;
    const size_t lds = (size_t)(9 * (g.Wb + 2) + g.Ws * C_o) * sizeof(float);
#define SW_C1(TT, COGV) hipLaunchKernelGGL((conv3d_wgrad_s2_c1_line_kernel<TT, COGV>), dim3((unsigned)lslots), block, lds, s, (const TT*)big, \
                                           (const TT*)small, workspace, N, g, C_o, lps)
    if (dtype == PYTC_BF16) { if (C_o == 32) SW_C1(bf16_t, 1); else SW_C1(bf16_t, 2); }
    else { if (C_o == 32) SW_C1(float, 1); else SW_C1(float, 2); }
#undef SW_C1
  } else if (C_k == 1 && g.kd == 3 && g.kh == 3 && g.kw == 3 && (dtype == PYTC_BF16 || dtype == PYTC_F32) &&
      tuning_get("conv3d_wgrad_thin", 1) != 0) {
    dim3 tgrid(slots, (C_o + SWT_TO - 1) / SWT_TO);
    if (dtype == PYTC_BF16)
      hipLaunchKernelGGL(conv3d_wgrad_strided_thin1_kernel<bf16_t>, tgrid, block, 0, s, (const bf16_t*)big, (const bf16_t*)small,
                         workspace, rows_total, g, C_o, rps, slots);
    else
      hipLaunchKernelGGL(conv3d_wgrad_strided_thin1_kernel<float>, tgrid, block, 0, s, (const float*)big, (const float*)small,
                         workspace, rows_total, g, C_o, rps, slots);
  } else if (C_k <= SWT_KMAX && (dtype == PYTC_BF16 || dtype == PYTC_F32) && tuning_get("conv3d_wgrad_thin", 1) != 0) {
    dim3 tgrid(slots, (C_o + SWT_TO - 1) / SWT_TO, taps);
    if (dtype == PYTC_BF16)
      hipLaunchKernelGGL(conv3d_wgrad_strided_thin_kernel<bf16_t>, tgrid, block, 0, s, (const bf16_t*)big, (const bf16_t*)small,
                         workspace, rows_total, g, C_k, C_o, rps, slots);
    else
      hipLaunchKernelGGL(conv3d_wgrad_strided_thin_kernel<float>, tgrid, block, 0, s, (const float*)big, (const float*)small,
                         workspace, rows_total, g, C_k, C_o, rps, slots);
  } else if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(conv3d_wgrad_strided_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)big, (const bf16_t*)small,
                       workspace, rows_total, g, C_k, C_o, rps, slots);
  else if (dtype == PYTC_F32)
    hipLaunchKernelGGL(conv3d_wgrad_strided_kernel<float>, grid, block, 0, s, (const float*)big, (const float*)small,
                       workspace, rows_total, g, C_k, C_o, rps, slots);
  else {
    set_error("conv3d_wgrad_strided: bad dtype %d", dtype);
    return PYTC_ERR_INVALID;
  }
  const long nW = (long)taps * C_o * C_k;
  hipLaunchKernelGGL(sw_reduce_slots_kernel, dim3(ceil_div(nW, 16)), dim3(256), 0, s, workspace, dW, nW, slots);
  PYTC_LAUNCH_CHECK("conv3d_wgrad_strided");
  return PYTC_OK;
}

/* ConvTranspose3d(k 3, s 2, p 1, output_padding 1) with ONE output channel, input-centric: workspace = 27 * N * Di * Hi * Wi floats */
extern "C" int pytc_convT3d_c1_fwd(const void* x, const float* w, const float* bias, void* y, float* workspace, int N,
                                   const int32_t* in_dims, int C_in, int dtype, void* stream) {
  PYTC_REQUIRE(x && w && y && workspace && in_dims && N >= 1, "convT3d_c1: bad arguments");
  PYTC_REQUIRE(C_in >= 8 && C_in % 8 == 0 && C_in <= 512, "convT3d_c1: C_in %% 8 == 0, <= 512 (got %d)", C_in);
  const long rows = (long)N * in_dims[0] * in_dims[1] * in_dims[2];
  const long total = rows * 8;
  const size_t lds = (size_t)27 * C_in * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PYTC_BF16) {
    hipLaunchKernelGGL(convT3d_c1_dots_kernel<bf16_t>, dim3((unsigned)((rows + 255) / 256)), dim3(256), lds, s, (const bf16_t*)x, w, workspace, rows, C_in);
    hipLaunchKernelGGL(convT3d_c1_gather_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, workspace, bias, (bf16_t*)y, N,
                       in_dims[0], in_dims[1], in_dims[2]);
  } else if (dtype == PYTC_F32) {
    hipLaunchKernelGGL(convT3d_c1_dots_kernel<float>, dim3((unsigned)((rows + 255) / 256)), dim3(256), lds, s, (const float*)x, w, workspace, rows, C_in);
    hipLaunchKernelGGL(convT3d_c1_gather_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, workspace, bias, (float*)y, N,
                       in_dims[0], in_dims[1], in_dims[2]);
  } else {
    PYTC_REQUIRE(false, "convT3d_c1: bad dtype %d", dtype);
  }
  PYTC_LAUNCH_CHECK("convT3d_c1");
  return PYTC_OK;
}

extern "C" int pytc_convT3d_thin_supported(int C_in, int C_out) {
  return (C_out >= 1 && C_out <= CT_OMAX && C_in >= 8 && C_in % 8 == 0 && (size_t)27 * C_out * C_in * 4 <= 64 * 1024) ? 1 : 0;
}

extern "C" int pytc_convT3d_thin_fwd(const void* x, const float* w, const float* bias, void* y, int N, const int32_t* in_dims,
                                     int C_in, int C_out, int dtype, void* stream) {
  PYTC_REQUIRE(x && w && y && in_dims && N >= 1, "convT3d_thin: bad arguments");
  PYTC_REQUIRE(pytc_convT3d_thin_supported(C_in, C_out), "convT3d_thin: needs C_out <= 4 and C_in %% 8 == 0 (got %d -> %d)", C_in, C_out);
  const long total = (long)N * 8 * in_dims[0] * in_dims[1] * in_dims[2];
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  const size_t lds = (size_t)27 * C_out * C_in * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(convT3d_thin_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), lds, s, (const bf16_t*)x, w, bias, (bf16_t*)y, N,
                       in_dims[0], in_dims[1], in_dims[2], C_in, C_out);
  else if (dtype == PYTC_F32)
    hipLaunchKernelGGL(convT3d_thin_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, s, (const float*)x, w, bias, (float*)y, N,
                       in_dims[0], in_dims[1], in_dims[2], C_in, C_out);
  else
    PYTC_REQUIRE(false, "convT3d_thin: bad dtype %d", dtype);
  PYTC_LAUNCH_CHECK("convT3d_thin");
  return PYTC_OK;
}
