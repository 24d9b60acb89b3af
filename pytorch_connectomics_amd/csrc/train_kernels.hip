// Backward kernels of the MedNeXt training step (correctness-first, two-stage deterministic reductions):
//   gelu fwd/bwd (elementwise), pointwise-conv weight gradient, depthwise-conv weight gradient (conv and
//   transposed conv share one form), GroupNorm backward (statistics + apply), strided depthwise backward-data,
//   elementwise add.  Data gradients of the 1x1 convs reuse pw_conv with transposed weights; the stride-1
//   depthwise backward-data reuses the forward kernels with flipped taps.
#include "pw_common.h"
#include "colstats.h"

namespace pytc {

__device__ __forceinline__ float gelu_grad(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  const float phi_big = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return phi_big + x * pdf;
}

template <typename T>
__global__ void __launch_bounds__(256)
gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float v = to_f32<T>(x[i]);
    out[i] = from_f32<T>(dy ? to_f32<T>(dy[i]) * gelu_grad(v) : gelu_erf(v));
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
add_kernel(T* __restrict__ y, const T* __restrict__ x, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = from_f32<T>(to_f32<T>(y[i]) + to_f32<T>(x[i]));
}

// copy of a channels-last (N, D, H, W, C) tensor with the FRONT faces (z, y or x == 0) zeroed: the output gradient an up block's mixer
// sees (those faces are the zero padding of its transposed conv, not outputs of the mixer); 16-byte pieces, one launch instead of
// clone + three strided fills
__global__ void __launch_bounds__(256)
copy_zero_front_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long pieces, int pieces_per_voxel, int D, int H, int W) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < pieces; i += stride) {
    const long v = i / pieces_per_voxel;
    const int x = (int)(v % W);
    const long q = v / W;
    const int y = (int)(q % H);
    const int z = (int)((q / H) % D);
    dst[i] = (x == 0 || y == 0 || z == 0) ? uint4{0u, 0u, 0u, 0u} : src[i];
  }
}

// ---- pointwise weight gradient: dWp[slot][o][k] = sum_{rows of slot} dY[r][o] * f(X[r][k]),  dbp[slot][o] ----------
// workgroup = one (row slot, 64x64 (o,k) tile); rows staged in LDS as fp32 32 at a time; thread = 4x4 (o,k) block.
constexpr int WG_TO = 64, WG_TK = 64, WG_TR = 32;
template <typename T>
__global__ void __launch_bounds__(256)
pw_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ ab, const T* __restrict__ dy,
                float* __restrict__ dWp, float* __restrict__ dbp, long rows_total, long rows_per_sample, int C_in,
                int C_out, long rows_per_slot, int slots, int x_act) {
  __shared__ float sx[WG_TR][WG_TK + 1];
  __shared__ float sd[WG_TR][WG_TO + 1];
  const int slot = blockIdx.x;
  const int tiles_k = (C_in + WG_TK - 1) / WG_TK;
  const int to = blockIdx.y / tiles_k, tk = blockIdx.y % tiles_k;
  const int o_base = to * WG_TO, k_base = tk * WG_TK;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;     // thread owns o = o_base + ty*4.., k = k_base + tx*4..
  float acc[4][4];
  float bacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const long r_begin = (long)slot * rows_per_slot;
  const long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  for (long r0 = r_begin; r0 < r_end; r0 += WG_TR) {
    for (int i = threadIdx.x; i < WG_TR * WG_TK; i += 256) {
      const int rr = i / WG_TK, kk = i % WG_TK;
      const long r = r0 + rr;
      float v = 0.f;
      if (r < r_end && k_base + kk < C_in) {
        v = to_f32<T>(x[r * C_in + k_base + kk]);
        if (ab) {
          const long n = r / rows_per_sample;
          v = fmaf(v, ab[(n * 2 + 0) * C_in + k_base + kk], ab[(n * 2 + 1) * C_in + k_base + kk]);
          v = to_f32<T>(from_f32<T>(v));      // the forward GEMM consumed the value rounded to T
        }
        if (x_act == PYTC_ACT_GELU) v = to_f32<T>(from_f32<T>(gelu_erf(v)));
      }
      sx[rr][kk] = v;
    }
    for (int i = threadIdx.x; i < WG_TR * WG_TO; i += 256) {
      const int rr = i / WG_TO, oo = i % WG_TO;
      const long r = r0 + rr;
      sd[rr][oo] = (r < r_end && o_base + oo < C_out) ? to_f32<T>(dy[r * C_out + o_base + oo]) : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < WG_TR; ++rr) {
      float dv[4], xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { dv[i] = sd[rr][ty * 4 + i]; xv[i] = sx[rr][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bacc[i] += dv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], xv[j], acc[i][j]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o_base + ty * 4 + i;
    if (o >= C_out) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k_base + tx * 4 + j;
      if (k < C_in) dWp[((long)slot * C_out + o) * C_in + k] = acc[i][j];
    }
    if (tk == 0 && tx == 0 && dbp) dbp[(long)slot * C_out + o] = bacc[i];
  }
}

// ---- pointwise weight gradient on MFMA (bf16): a "TN" GEMM with the voxel rows as the reduction dimension ----------
// Both operands are [row][channel] in HBM (channels contiguous) while an MFMA fragment wants 8 reduction indices of
// ONE channel per lane: gfx950's LDS transpose read (ds_read_b64_tr_b16) does that regrouping.  Every wave owns 32
// rows at a time: 16-B global loads -> a wave-private, row-padded LDS image (no workgroup barrier in the loop, the next
// block's loads are in flight during the MFMAs) -> per 16-channel tile two transpose reads = one fragment.  The
// row <-> k-slot map (lane group g, element j) -> row (j>>2)*16 + g*4 + (j&3) is the same for both operands, which is
// all a reduction needs; with a row pitch of 2*C+32 bytes each half-wave's 8 rows fall in 8 distinct bank octets.
// db comes from one extra MFMA per tile row against a fragment of ones.  The four waves' accumulators are added in a
// fixed order (wave 0 + 1 + 2 + 3) -> per-slot partials -> reduce_slots: deterministic.
// DG (round 5): the same pass ALSO forms the data gradient of the projecting conv behind the activation,
//     dxo[r][k] = bf16( (sum_o W[o][k] dy[r][o]) * gelu'(x[r][k]) ),      x = the hidden pre-activation, W = the conv's [C_out][C_in] weight,
// from the rows it has staged anyway: the gradient rows are the B operand of one more MFMA per 16 x 16 tile (A = the paired image of
// W^T, read once per wave), the pre-activation comes back from a second copy of the staged rows (the first is turned into gelu(x) for the
// weight gradient), and the product leaves as 16-byte stores.  One pass over (x, dy) instead of two -- pytc_pw_wgrad reads them, then
// pytc_pw_conv_fwd(RES_GELU_BWD) reads them again: 3 x 64 B per voxel saved per level-0 block -- the weight gradient with the same bits as pytc_pw_wgrad, dx equal to that GEMM on the paired weight image up to one bf16 ulp in a few
// outputs per million (same MFMA on the same operands; the GELU' expression is contracted differently inside the two kernels -- measured
// 2 of 529 024 and 13 of 12.8 M elements).  Needs the whole channel extent in one workgroup: C_out = 16 MT,
// C_in = 16 NT, gridDim.y = 1.
template <int MT, int NT, bool DG = false>
__global__ void __launch_bounds__(256, (MT * NT >= 8 ? 2 : (MT * NT >= 4 ? 3 : 4)))
pw_wgrad_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ab, const bf16_t* __restrict__ dy,
                     float* __restrict__ dWp, float* __restrict__ dbp, long rows_total, long rows_per_sample, int C_in,
                     int C_out, long rows_per_slot, int x_act, int sps, int ab_mode, const bf16x8_t* __restrict__ w_dg = nullptr,
                     bf16_t* __restrict__ dxo = nullptr) {
  // sps > 0: `sps` slots PER SAMPLE (slot = n * sps + j covers rows of sample n only): the partials then are per-sample sums,
  // which pytc_pw_wgrad_groupnorm turns into the GroupNorm backward statistics.  ab_mode 1: `ab` holds (mean, rstd) and the
  // operand is the normalised xhat = (x - mean) * rstd instead of a * x + b.
  constexpr int BM = MT * 16, BN = NT * 16;
  constexpr int SG = BM * 2 + 32, SX = BN * 2 + 32;        // LDS row pitch in bytes
  constexpr int WAVE_BYTES = 32 * (SG + 2 * SX);             // dY rows, operand rows, and the operand's low part (ab_mode 1)
  constexpr int RED_BYTES = (BM * BN + BM) * 4;
  constexpr int LDS_BYTES = 4 * WAVE_BYTES > RED_BYTES ? 4 * WAVE_BYTES : RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* lg = lds + wave * WAVE_BYTES;
  unsigned char* lx = lg + 32 * SG;
  unsigned char* lx2 = lx + 32 * SX;
  // ab_mode 1 (xhat operand): the operand is split into bf16 high and low parts, two MFMAs per tile -- xhat to 2^-17 instead of
  // 2^-9, because the sums of this mode become the statistics of a norm backward whose apply pass sees the exact xhat
  const bool split = ab_mode != 0;
  const int slot = blockIdx.x;
  const int tiles_k = C_in / BN;
  const int o_base = (blockIdx.y / tiles_k) * BM, k_base = (blockIdx.y % tiles_k) * BN;
  const bool want_db = dbp != nullptr && (blockIdx.y % tiles_k) == 0;
  long r_begin = (long)slot * rows_per_slot;
  long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  if (sps > 0) {
    const long sample_begin = (long)(slot / sps) * rows_per_sample;
    r_begin = sample_begin + (long)(slot % sps) * rows_per_slot;
    r_end = r_begin + rows_per_slot < sample_begin + rows_per_sample ? r_begin + rows_per_slot : sample_begin + rows_per_sample;
  }

  constexpr int CHG = BM / 8, RG = 64 / CHG, ITG = 32 / RG;   // 16-B chunks per row, rows per load, loads per block
  constexpr int CHX = BN / 8, RX = 64 / CHX, ITX = 32 / RX;
  const int g_row = lane / CHG, g_chunk = lane % CHG;
  const int x_row = lane / CHX, x_chunk = lane % CHX;
  // two register sets: the loads of TWO row blocks are in flight while one is multiplied (one block ahead left each wave waiting
  // on HBM for most of an iteration: 12 MFMAs cover ~200 cycles of a ~2 us round trip; measured 2.3 TB/s at level 0)
  typedef unsigned int q4_t __attribute__((ext_vector_type(4)));      // (a uint4 STRUCT array ends up in scratch memory here)
  q4_t rgA[ITG], rxA[ITX], rgB[ITG], rxB[ITX];
  const q4_t zero4 = {0u, 0u, 0u, 0u};
  auto fetch = [&](long r0, q4_t (&rg)[ITG], q4_t (&rx)[ITX]) {
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const long r = r0 + it * RG + g_row;
      rg[it] = r < r_end ? *reinterpret_cast<const q4_t*>(dy + r * C_out + o_base + g_chunk * 8) : zero4;
    }
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
      const long r = r0 + it * RX + x_row;
      rx[it] = r < r_end ? *reinterpret_cast<const q4_t*>(x + r * C_in + k_base + x_chunk * 8) : zero4;
    }
  };
  // norm affine of the lane's 8 channels, cached per sample: a 32-row block never straddles two samples when
  // rows_per_sample % 32 == 0 (every MedNeXt level at 112^3), so the sample index is tracked per wave with a
  // compare instead of a 64-bit division per row, and (a, b) are reloaded only when it changes
  // (... and every slot starts on a multiple of 32 rows: with rows_per_slot = 8232 and three samples of 112^3 rows the block at a sample
  // boundary took the previous sample's affine for up to 31 rows -- found in round 6, tests/test_gpu_training.py
  // test_pw_wgrad_mfma_affine_at_sample_boundaries)
  const bool uniform_n = ab != nullptr && (sps > 0 || ((rows_per_sample % 32) == 0 && (rows_per_slot % 32) == 0));
  float av[8], bv[8];
  long n_cached = -1;
  auto load_ab = [&](long n) {
    const float* a = ab + (n * 2 + 0) * C_in + k_base + x_chunk * 8;
    const float* b = ab + (n * 2 + 1) * C_in + k_base + x_chunk * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (ab_mode) { av[i] = b[i]; bv[i] = -a[i] * b[i]; }       // (mean, rstd) -> xhat = rstd * x - mean * rstd
      else { av[i] = a[i]; bv[i] = b[i]; }
    }
    n_cached = n;
  };
  auto stage = [&](long r0, q4_t (&rg)[ITG], q4_t (&rx)[ITX]) {
#pragma unroll
    for (int it = 0; it < ITG; ++it) *reinterpret_cast<q4_t*>(lg + (it * RG + g_row) * SG + g_chunk * 16) = rg[it];
    if (uniform_n) {
      const long n = r0 / rows_per_sample;            // r0 is wave-uniform: one division per 32-row block
      if (n != n_cached) load_ab(n);
    }
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
      q4_t v = rx[it];
      const long r = r0 + it * RX + x_row;
      if (ab && r < r_end) {            // the forward GEMM consumed bf16(a*x+b): restate it here
        if (!uniform_n) load_ab(r / rows_per_sample);
        const bf16x8_t in = __builtin_bit_cast(bf16x8_t, v);
        f32x8_t f = __builtin_convertvector(in, f32x8_t);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], av[i], bv[i]);
        const bf16x8_t hi = __builtin_convertvector(f, bf16x8_t);
        v = __builtin_bit_cast(q4_t, hi);
        if (split) {
          const f32x8_t back = __builtin_convertvector(hi, f32x8_t);
          *reinterpret_cast<q4_t*>(lx2 + (it * RX + x_row) * SX + x_chunk * 16) =
              __builtin_bit_cast(q4_t, __builtin_convertvector(f - back, bf16x8_t));
        }
      } else if (split) {
        *reinterpret_cast<q4_t*>(lx2 + (it * RX + x_row) * SX + x_chunk * 16) = zero4;
      }
      if constexpr (DG) *reinterpret_cast<q4_t*>(lx2 + (it * RX + x_row) * SX + x_chunk * 16) = v;     // the raw pre-activation rows (split is off)
      if (x_act == PYTC_ACT_GELU) {     // the forward GEMM consumed bf16(gelu(x)) (fused pre-activation)
        f32x8_t f = __builtin_convertvector(__builtin_bit_cast(bf16x8_t, v), f32x8_t);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = gelu_fast(f[i]);      // the function pw_fast's prologue applied in the forward
        v = __builtin_bit_cast(q4_t, __builtin_convertvector(f, bf16x8_t));
      }
      *reinterpret_cast<q4_t*>(lx + (it * RX + x_row) * SX + x_chunk * 16) = v;
    }
  };
  // transpose read of one 16-channel fragment: lane (g = lane>>4, i = lane&15) supplies the address of 4 channels
  // (i&3)*4.. of row g*4 + (i>>2) (+16 for the second read) and receives channel i of the group's 4 rows.
  const int fr_row = (lane >> 4) * 4 + ((lane & 15) >> 2), fr_col = (lane & 3) * 8;
  auto frag = [&](unsigned char* base, int pitch, int tile) -> bf16x8_t {
    unsigned char* p = base + fr_row * pitch + tile * 32 + fr_col;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 16 * pitch));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, both);
  };

  f32x4_t acc[MT][NT], accb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    accb[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8_t ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

  // DG: A fragments of W^T (paired image [C_in][C_out]: NT M-tiles x one 32-wide k-step per MT = 2), constant for the wave
  bf16x8_t wdg[DG ? NT : 1];
  if constexpr (DG) {
    static_assert(!DG || MT == 2, "the fused data gradient is written for C_out = 32 (one k-step)");
#pragma unroll
    for (int t = 0; t < NT; ++t) wdg[t] = w_dg[t * 64 + lane];
  }
  auto multiply = [&](long r0) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if constexpr (DG) {
      const int rr = lane & 15, kb = lane >> 4;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {                  // the two 16-row tiles of the 32-row block
        const bf16x8_t bdy = *reinterpret_cast<const bf16x8_t*>(lg + (t2 * 16 + rr) * SG + kb * 16);
        const long row = r0 + t2 * 16 + rr;
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
          const f32x4_t lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdg[2 * pr], bdy, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          const f32x4_t hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdg[2 * pr + 1], bdy, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          float hv[8], v[8];
          VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(lx2 + (t2 * 16 + rr) * SX + (pr * 32 + kb * 8) * 2), hv);
#pragma unroll
          for (int i = 0; i < 4; ++i) { v[i] = lo[i] * gelu_erf_grad(hv[i]); v[4 + i] = hi[i] * gelu_erf_grad(hv[4 + i]); }
          if (row < r_end) VecIO<bf16_t, 8>::store(dxo + row * C_in + pr * 32 + kb * 8, v);
        }
      }
    }
    bf16x8_t fa[MT], fb[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) fa[m] = frag(lg, SG, m);
#pragma unroll
    for (int n = 0; n < NT; ++n) fb[n] = frag(lx, SX, n);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
      if (want_db) accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], ones, accb[m], 0, 0, 0);
    }
    if (split) {
#pragma unroll
      for (int n = 0; n < NT; ++n) fb[n] = frag(lx2, SX, n);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  };
  long r0 = r_begin + wave * 32;
  if (r0 < r_end) fetch(r0, rgA, rxA);
  if (r0 + 128 < r_end) fetch(r0 + 128, rgB, rxB);
  for (; r0 < r_end; r0 += 256) {                    // same block order per wave as before: the sums are bit-identical
    stage(r0, rgA, rxA);
    if (r0 + 256 < r_end) fetch(r0 + 256, rgA, rxA);
    multiply(r0);
    if (r0 + 128 < r_end) {
      stage(r0 + 128, rgB, rxB);
      if (r0 + 384 < r_end) fetch(r0 + 384, rgB, rxB);
      multiply(r0 + 128);
    }
  }

  // fixed-order cross-wave sum through LDS, then wave 0 stores the slot partial
  float* red = reinterpret_cast<float*>(lds);
  const int nn = lane & 15, mg = (lane >> 4) * 4;
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) red[(m * 16 + mg + i) * BN + n * 16 + nn] = acc[m][n][i];
        if (nn == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) red[BM * BN + m * 16 + mg + i] = accb[m][i];
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[m][n][i] += red[(m * 16 + mg + i) * BN + n * 16 + nn];
        if (nn == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) accb[m][i] += red[BM * BN + m * 16 + mg + i];
        }
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = o_base + m * 16 + mg + i;
#pragma unroll
        for (int n = 0; n < NT; ++n) dWp[((long)slot * C_out + o) * C_in + k_base + n * 16 + nn] = acc[m][n][i];
        if (want_db && nn == 0) dbp[(long)slot * C_out + o] = accb[m][i];
      }
    }
  }
}

// ---- level-0 mixer backward with the hidden pre-activation REBUILT from the depthwise output (round 6) ------------------------------
// A block's training forward stored the hidden pre-activation hp = bf16(W2 bf16(a t + b) + b2) (2 C_hid bytes per voxel written, read
// back once by the pass above).  At the full-resolution level, where every kernel is bound by its bytes, this kernel takes the
// depthwise output t instead and rebuilds hp in registers with the forward's own arithmetic -- the paired image of W2 as the A operand,
// the lane's 8 normalised channels of a row as the B operand, accumulators started from b2: one 16x16x32 MFMA per tile, bit-identical to
// what pw_mlp_kernel<.., STOREH> rounded -- so the forward stores nothing (pytc_pw_mlp_train_fwd with a null hidden buffer) and the
// backward reads 64 B per voxel where it read 128.  Per 32-row block of a wave:
//   stage     dy rows -> wave-private LDS image (for the transposed weight-gradient fragments); t rows -> bf16(a t + b) in registers
//             (GN: also xhat = (t - mean) rstd split into bf16 high and low parts -> two LDS images)
//   rebuild   hp tile pair (lo, hi) = 2 MFMAs; g = bf16(gelu_fast(hp)) -> LDS image (the weight gradient's operand);
//             dhp = bf16((W3^T dy) * gelu_fast'(hp)) = 2 more MFMAs on the dy rows still in registers -> 16-byte stores (the derivative of
//             the function the forward evaluated, from the same exponential and reciprocal: gelu_fast_with_grad)
//   dW3 += dy^T g, db3 += dy^T 1 through ds_read_tr16_b64 fragments, exactly as pw_wgrad_mfma_kernel
//   GN        the dhp rows (kept in registers) replace g in its LDS image; dW2-partials M += dhp^T xhat_hi + dhp^T xhat_lo,
//             q += dhp^T 1: the per-sample sums that norm_bwd_from_wgrad_kernel turns into the GroupNorm backward statistics, the
//             norm-backward coefficients and dW2 / db2 -- the pass of pytc_pw_wgrad_groupnorm over (t, dhp) is not run at all.
// Slots never straddle samples (`sps` per sample, as pytc_pw_wgrad_groupnorm), so the affine of a workgroup is one sample's.  With the
// slot count of the launches it replaces the partial sums group the same rows in the same order: dW3 / db3 / M / q carry the bits of
// pytc_pw_wgrad_dgrad_partial / pytc_pw_wgrad_groupnorm whenever those launches' slots do not straddle samples either.
// C = C_out = 32, C_hid = 16 HT (HT even).
struct MixBwd {
  const bf16_t* t;          // [N][rows][32]
  const float* ab;          // [N][2][32]
  const float* mr;          // [N][2][32] (mean, rstd), GN
  const bf16_t* dy;         // [N][rows][32]
  const bf16x8_t* w2p;      // paired image of W2 (32 -> C_hid): HT tiles x 64 lanes
  const float* b2;          // [C_hid]
  const bf16x8_t* w3t;      // paired image of W3^T ([C_hid][32]): HT tiles x 64 lanes
  float* dW3p; float* db3p; // [slots][32][C_hid], [slots][32]
  float* dW2p; float* db2p; // GN: [slots][C_hid][32], [slots][C_hid]
  bf16_t* dhp;              // [N][rows][C_hid]
  long rows_per_sample, rows_per_slot;
  int sps, want_db3;
  int probe;                // measurements only (wrong results): 1 = identity in the place of the activation and its derivative
};

template <int HT, bool GN>
__global__ void __launch_bounds__(256, 2)
mixer_bwd_rc_kernel(MixBwd p) {
  static_assert(HT % 2 == 0 && HT >= 2 && HT <= 8, "hidden tiles come in pairs");
  constexpr int CH = HT * 16;
  constexpr int SG = 32 * 2 + 32, SX = CH * 2 + 32;            // LDS row pitches (bytes) of the 32-channel and the hidden images
  constexpr int WAVE_BYTES = 32 * (SG + SX + (GN ? 2 * SG : 0));
  constexpr int RED3 = (32 * CH + 32) * 4, RED2 = GN ? (CH * 32 + CH) * 4 : 0;
  constexpr int TAB_BYTES = (CH + (GN ? 64 : 0)) * 4;           // b2 | xhat scale | xhat offset
  constexpr int LDS_MAIN = 4 * WAVE_BYTES > RED3 + RED2 ? 4 * WAVE_BYTES : RED3 + RED2;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_MAIN + TAB_BYTES];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  typedef unsigned int q4_t __attribute__((ext_vector_type(4)));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rr = lane & 15, kb = lane >> 4;
  unsigned char* lg = lds + wave * WAVE_BYTES;      // dy rows
  unsigned char* lx = lg + 32 * SG;                 // gelu(hp) rows, then (GN) dhp rows
  unsigned char* lxh = lx + 32 * SX;                // GN: xhat high / low parts
  unsigned char* lxl = lxh + 32 * SG;
  float* tab = reinterpret_cast<float*>(lds + LDS_MAIN);

  const int slot = blockIdx.x;
  const long n = slot / p.sps;
  const long sample_begin = n * p.rows_per_sample;
  const long r_begin = sample_begin + (long)(slot % p.sps) * p.rows_per_slot;
  const long r_end = r_begin + p.rows_per_slot < sample_begin + p.rows_per_sample ? r_begin + p.rows_per_slot : sample_begin + p.rows_per_sample;

  for (int i = threadIdx.x; i < CH; i += 256) tab[i] = p.b2[i];
  if constexpr (GN) {
    if (threadIdx.x < 32) {
      const float mean = p.mr[(n * 2 + 0) * 32 + threadIdx.x], rstd = p.mr[(n * 2 + 1) * 32 + threadIdx.x];
      tab[CH + threadIdx.x] = rstd;
      tab[CH + 32 + threadIdx.x] = -mean * rstd;
    }
  }
  // the norm affine of the lane's 8 channels (one sample per workgroup)
  float av[8], bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { av[i] = p.ab[(n * 2 + 0) * 32 + kb * 8 + i]; bv[i] = p.ab[(n * 2 + 1) * 32 + kb * 8 + i]; }
  bf16x8_t w2f[HT], w3f[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) { w2f[t] = p.w2p[t * 64 + lane]; w3f[t] = p.w3t[t * 64 + lane]; }
  __syncthreads();

  const q4_t zero4 = {0u, 0u, 0u, 0u};
  q4_t rtA[2], rgA[2], rtB[2], rgB[2];
  auto fetch = [&](long r0, q4_t (&rt)[2], q4_t (&rg)[2]) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const long r = r0 + t2 * 16 + rr;
      rt[t2] = r < r_end ? *reinterpret_cast<const q4_t*>(p.t + r * 32 + kb * 8) : zero4;
      rg[t2] = r < r_end ? *reinterpret_cast<const q4_t*>(p.dy + r * 32 + kb * 8) : zero4;
    }
  };
  const int fr_row = (lane >> 4) * 4 + ((lane & 15) >> 2), fr_col = (lane & 3) * 8;
  auto frag = [&](unsigned char* base, int pitch, int tile) -> bf16x8_t {
    unsigned char* q = base + fr_row * pitch + tile * 32 + fr_col;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)q);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + 16 * pitch));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, both);
  };

  f32x4_t acc3[2][HT], accb3[2];
  f32x4_t acc2[GN ? HT : 1][2], accb2[GN ? HT : 1];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    accb3[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < HT; ++h) acc3[m][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int h = 0; h < (GN ? HT : 1); ++h) {
    accb2[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    acc2[h][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    acc2[h][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8_t ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

  bf16x8_t tn[2], dyb[2];
  auto stage = [&](long r0, q4_t (&rt)[2], q4_t (&rg)[2]) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      dyb[t2] = __builtin_bit_cast(bf16x8_t, rg[t2]);
      *reinterpret_cast<q4_t*>(lg + (t2 * 16 + rr) * SG + kb * 16) = rg[t2];
      const f32x8_t raw = __builtin_convertvector(__builtin_bit_cast(bf16x8_t, rt[t2]), f32x8_t);
      f32x8_t f;
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = fmaf(raw[i], av[i], bv[i]);
      tn[t2] = __builtin_convertvector(f, bf16x8_t);          // what the forward GEMM consumed
      if constexpr (GN) {
        const bool live = r0 + t2 * 16 + rr < r_end;
        const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(tab + CH + kb * 8), s1 = *reinterpret_cast<const f32x4_t*>(tab + CH + kb * 8 + 4);
        const f32x4_t o0 = *reinterpret_cast<const f32x4_t*>(tab + CH + 32 + kb * 8), o1 = *reinterpret_cast<const f32x4_t*>(tab + CH + 32 + kb * 8 + 4);
        f32x8_t xh;
#pragma unroll
        for (int i = 0; i < 4; ++i) { xh[i] = fmaf(raw[i], s0[i], o0[i]); xh[4 + i] = fmaf(raw[4 + i], s1[i], o1[i]); }
        const bf16x8_t hi = __builtin_convertvector(xh, bf16x8_t);
        const f32x8_t back = __builtin_convertvector(hi, f32x8_t);
        const bf16x8_t lo = __builtin_convertvector(xh - back, bf16x8_t);
        *reinterpret_cast<q4_t*>(lxh + (t2 * 16 + rr) * SG + kb * 16) = live ? __builtin_bit_cast(q4_t, hi) : zero4;
        *reinterpret_cast<q4_t*>(lxl + (t2 * 16 + rr) * SG + kb * 16) = live ? __builtin_bit_cast(q4_t, lo) : zero4;
      }
    }
  };
  auto compute = [&](long r0) {
    q4_t dq[GN ? 2 : 1][GN ? HT / 2 : 1];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const long row = r0 + t2 * 16 + rr;
#pragma unroll
      for (int pr = 0; pr < HT / 2; ++pr) {
        const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(tab + pr * 32 + kb * 8), c1 = *reinterpret_cast<const f32x4_t*>(tab + pr * 32 + kb * 8 + 4);
        const f32x4_t lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[2 * pr], tn[t2], c0, 0, 0, 0);
        const f32x4_t hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[2 * pr + 1], tn[t2], c1, 0, 0, 0);
        float pre[8], hv[8], g[8], gd[8], v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { pre[i] = lo[i]; pre[4 + i] = hi[i]; }
        const bf16x8_t hb = Mma<bf16_t>::from_floats(pre);      // the value the forward rounded (and used to store)
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&hb), hv);
        // g = gelu_fast(hp) with gelu_fast's bits; g' = the derivative of that function (pytc_common.h: <= 1.1e-4 from the erf form's)
#pragma unroll
        for (int i = 0; i < 8; ++i) gelu_fast_with_grad(hv[i], g[i], gd[i]);
        if (p.probe == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { g[i] = hv[i]; gd[i] = 1.0f; }
        }
        *reinterpret_cast<bf16x8_t*>(lx + (t2 * 16 + rr) * SX + (pr * 32 + kb * 8) * 2) = Mma<bf16_t>::from_floats(g);
        const f32x4_t dlo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[2 * pr], dyb[t2], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const f32x4_t dhi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[2 * pr + 1], dyb[t2], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = dlo[i] * gd[i]; v[4 + i] = dhi[i] * gd[4 + i]; }
        const bf16x8_t vb = Mma<bf16_t>::from_floats(v);
        if (row < r_end) *reinterpret_cast<bf16x8_t*>(p.dhp + row * CH + pr * 32 + kb * 8) = vb;
        if constexpr (GN) dq[t2][pr] = row < r_end ? __builtin_bit_cast(q4_t, vb) : zero4;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    bf16x8_t fa[2], fb[HT];
#pragma unroll
    for (int m = 0; m < 2; ++m) fa[m] = frag(lg, SG, m);
#pragma unroll
    for (int h = 0; h < HT; ++h) fb[h] = frag(lx, SX, h);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int h = 0; h < HT; ++h) acc3[m][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[h], acc3[m][h], 0, 0, 0);
      if (p.want_db3) accb3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], ones, accb3[m], 0, 0, 0);
    }
    if constexpr (GN) {
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int pr = 0; pr < HT / 2; ++pr) *reinterpret_cast<q4_t*>(lx + (t2 * 16 + rr) * SX + (pr * 32 + kb * 8) * 2) = dq[t2][pr];
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      bf16x8_t fh[2], fl[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) { fh[c] = frag(lxh, SG, c); fl[c] = frag(lxl, SG, c); }
#pragma unroll
      for (int h = 0; h < HT; ++h) {
        const bf16x8_t fd = frag(lx, SX, h);
#pragma unroll
        for (int c = 0; c < 2; ++c) acc2[h][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd, fh[c], acc2[h][c], 0, 0, 0);
        accb2[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd, ones, accb2[h], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 2; ++c) acc2[h][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd, fl[c], acc2[h][c], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  };

  long r0 = r_begin + wave * 32;
  if (r0 < r_end) fetch(r0, rtA, rgA);
  if (r0 + 128 < r_end) fetch(r0 + 128, rtB, rgB);
  for (; r0 < r_end; r0 += 256) {                    // the block order per wave of pw_wgrad_mfma_kernel
    stage(r0, rtA, rgA);
    if (r0 + 256 < r_end) fetch(r0 + 256, rtA, rgA);
    compute(r0);
    if (r0 + 128 < r_end) {
      stage(r0 + 128, rtB, rgB);
      if (r0 + 384 < r_end) fetch(r0 + 384, rtB, rgB);
      compute(r0 + 128);
    }
  }

  // fixed-order cross-wave sums through LDS (wave 0 + 1 + 2 + 3), then wave 0 stores the slot partials
  __syncthreads();
  float* red3 = reinterpret_cast<float*>(lds);
  float* red2 = reinterpret_cast<float*>(lds + RED3);
  const int nn = lane & 15, mg = (lane >> 4) * 4;
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) red3[(m * 16 + mg + i) * CH + h * 16 + nn] = acc3[m][h][i];
        if (nn == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) red3[32 * CH + m * 16 + mg + i] = accb3[m][i];
        }
      }
      if constexpr (GN) {
#pragma unroll
        for (int h = 0; h < HT; ++h) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) red2[(h * 16 + mg + i) * 32 + c * 16 + nn] = acc2[h][c][i];
          if (nn == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red2[CH * 32 + h * 16 + mg + i] = accb2[h][i];
          }
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc3[m][h][i] += red3[(m * 16 + mg + i) * CH + h * 16 + nn];
        if (nn == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) accb3[m][i] += red3[32 * CH + m * 16 + mg + i];
        }
      }
      if constexpr (GN) {
#pragma unroll
        for (int h = 0; h < HT; ++h) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[h][c][i] += red2[(h * 16 + mg + i) * 32 + c * 16 + nn];
          if (nn == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) accb2[h][i] += red2[CH * 32 + h * 16 + mg + i];
          }
        }
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = m * 16 + mg + i;
#pragma unroll
        for (int h = 0; h < HT; ++h) p.dW3p[((long)slot * 32 + o) * CH + h * 16 + nn] = acc3[m][h][i];
        if (p.want_db3 && nn == 0) p.db3p[(long)slot * 32 + o] = accb3[m][i];
      }
    }
    if constexpr (GN) {
#pragma unroll
      for (int h = 0; h < HT; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int o = h * 16 + mg + i;
#pragma unroll
          for (int c = 0; c < 2; ++c) p.dW2p[((long)slot * CH + o) * 32 + c * 16 + nn] = acc2[h][c][i];
          if (nn == 0) p.db2p[(long)slot * CH + o] = accb2[h][i];
        }
      }
    }
  }
}

// ---- the projecting conv's weight gradient + data gradient in one pass, WIDE hidden layer (round 6; the 64 -> 128 -> 32 up block) -----------
// pw_wgrad_mfma_kernel<2, NT, true> keeps two LDS copies of the hidden rows per wave (gelu(x) for the weight gradient, x for GELU'): at 128
// hidden channels 21.5 KB per wave, one workgroup per CU.  Here the lane's global loads ARE the MFMA result layout of the data gradient --
// lane (row r, group kb) loads the 8 hidden channels pr * 32 + kb * 8 .. of its row for every tile pair pr -- so the pre-activation meets
// (W^T dy) in registers, and only gelu(x) (weight-gradient operand) and the dy rows go through LDS (12.3 KB per wave).  One pass over
// (x, dy) instead of pytc_pw_wgrad + pytc_pw_conv_fwd(RES_GELU_BWD): 3.2 GB instead of 5.0 GB per 4 x 112^3 step at 128 hidden channels.
// dW / db: the rows per slot and the block order per wave of pytc_pw_wgrad_partial -> the same bits.  dx = bf16((W^T dy) * g'(x)) with g' the
// derivative of gelu_fast (the function the training forward evaluated; gelu_fast_with_grad: <= 1.1e-4 from the erf form's).  C_out = 32.
template <int HT>
__global__ void __launch_bounds__(256, 2)
pw_wgrad_dgrad_wide_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const bf16x8_t* __restrict__ w_dg,
                           float* __restrict__ dWp, float* __restrict__ dbp, bf16_t* __restrict__ dxo, long rows_total, long rows_per_slot) {
  static_assert(HT % 2 == 0 && HT >= 2 && HT <= 8, "hidden tiles come in pairs");
  constexpr int CH = HT * 16;
  constexpr int SG = 32 * 2 + 32, SX = CH * 2 + 32;
  constexpr int WAVE_BYTES = 32 * (SG + SX);
  constexpr int RED = (32 * CH + 32) * 4;
  constexpr int LDS_BYTES = 4 * WAVE_BYTES > RED ? 4 * WAVE_BYTES : RED;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  typedef unsigned int q4_t __attribute__((ext_vector_type(4)));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rr = lane & 15, kb = lane >> 4;
  unsigned char* lg = lds + wave * WAVE_BYTES;
  unsigned char* lx = lg + 32 * SG;
  const int slot = blockIdx.x;
  const long r_begin = (long)slot * rows_per_slot;
  const long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  const bool want_db = dbp != nullptr;

  bf16x8_t wf[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) wf[t] = w_dg[t * 64 + lane];
  const q4_t zero4 = {0u, 0u, 0u, 0u};
  q4_t rhA[2][HT / 2], rgA[2], rhB[2][HT / 2], rgB[2];
  auto fetch = [&](long r0, q4_t (&rh)[2][HT / 2], q4_t (&rg)[2]) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const long r = r0 + t2 * 16 + rr;
      rg[t2] = r < r_end ? *reinterpret_cast<const q4_t*>(dy + r * 32 + kb * 8) : zero4;
#pragma unroll
      for (int pr = 0; pr < HT / 2; ++pr)
        rh[t2][pr] = r < r_end ? *reinterpret_cast<const q4_t*>(x + r * CH + pr * 32 + kb * 8) : zero4;
    }
  };
  const int fr_row = (lane >> 4) * 4 + ((lane & 15) >> 2), fr_col = (lane & 3) * 8;
  auto frag = [&](unsigned char* base, int pitch, int tile) -> bf16x8_t {
    unsigned char* q = base + fr_row * pitch + tile * 32 + fr_col;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)q);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + 16 * pitch));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, both);
  };
  f32x4_t acc[2][HT], accb[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    accb[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < HT; ++h) acc[m][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8_t ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

  auto block = [&](long r0, q4_t (&rh)[2][HT / 2], q4_t (&rg)[2]) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const long row = r0 + t2 * 16 + rr;
      *reinterpret_cast<q4_t*>(lg + (t2 * 16 + rr) * SG + kb * 16) = rg[t2];
      const bf16x8_t dyb = __builtin_bit_cast(bf16x8_t, rg[t2]);
#pragma unroll
      for (int pr = 0; pr < HT / 2; ++pr) {
        float hv[8], g[8], gd[8], v[8];
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rh[t2][pr]), hv);
#pragma unroll
        for (int i = 0; i < 8; ++i) gelu_fast_with_grad(hv[i], g[i], gd[i]);
        *reinterpret_cast<bf16x8_t*>(lx + (t2 * 16 + rr) * SX + (pr * 32 + kb * 8) * 2) = Mma<bf16_t>::from_floats(g);
        const f32x4_t dlo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * pr], dyb, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const f32x4_t dhi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * pr + 1], dyb, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = dlo[i] * gd[i]; v[4 + i] = dhi[i] * gd[4 + i]; }
        if (row < r_end) *reinterpret_cast<bf16x8_t*>(dxo + row * CH + pr * 32 + kb * 8) = Mma<bf16_t>::from_floats(v);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    bf16x8_t fa[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) fa[m] = frag(lg, SG, m);
#pragma unroll
    for (int h = 0; h < HT; ++h) {
      const bf16x8_t fb = frag(lx, SX, h);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb, acc[m][h], 0, 0, 0);
    }
    if (want_db) {
#pragma unroll
      for (int m = 0; m < 2; ++m) accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], ones, accb[m], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  };
  // a set's next loads are issued right after its block and land during the other set's block (the hidden rows stay in their registers
  // through the block: no second copy)
  long r0 = r_begin + wave * 32;
  if (r0 < r_end) fetch(r0, rhA, rgA);
  if (r0 + 128 < r_end) fetch(r0 + 128, rhB, rgB);
  for (; r0 < r_end; r0 += 256) {                    // the block order per wave of pw_wgrad_mfma_kernel
    block(r0, rhA, rgA);
    if (r0 + 256 < r_end) fetch(r0 + 256, rhA, rgA);
    if (r0 + 128 < r_end) {
      block(r0 + 128, rhB, rgB);
      if (r0 + 384 < r_end) fetch(r0 + 384, rhB, rgB);
    }
  }
  float* red = reinterpret_cast<float*>(lds);
  const int nn = lane & 15, mg = (lane >> 4) * 4;
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) red[(m * 16 + mg + i) * CH + h * 16 + nn] = acc[m][h][i];
        if (nn == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) red[32 * CH + m * 16 + mg + i] = accb[m][i];
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[m][h][i] += red[(m * 16 + mg + i) * CH + h * 16 + nn];
        if (nn == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) accb[m][i] += red[32 * CH + m * 16 + mg + i];
        }
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = m * 16 + mg + i;
#pragma unroll
        for (int h = 0; h < HT; ++h) dWp[((long)slot * 32 + o) * CH + h * 16 + nn] = acc[m][h][i];
        if (want_db && nn == 0) dbp[(long)slot * 32 + o] = accb[m][i];
      }
    }
  }
}

// ---- thin pointwise weight gradient (C_in == 1: stem, or C_out == 1: one-channel heads; no prologue) --------------
// dW[c] = sum_r big[r][c] * thin[r]: the column-sum lane map of colstats.h (16-byte loads of the wide operand).
// big_is_dy: big = dY [rows][C_out], thin = X [rows][1], db[c] = sum big;  else big = X [rows][C_in], thin = dY, db[0].
template <typename T>
__global__ void __launch_bounds__(256)
pw_wgrad_thin_kernel(const T* __restrict__ big, const T* __restrict__ thin, float* __restrict__ dWp,
                     float* __restrict__ dbp, long rows_total, int C, long rows_per_slot, int big_is_dy) {
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ float lds[2 * 256 * EPV];
  const int slot = blockIdx.x;
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < rows_total ? r0 + rows_per_slot : rows_total;
  const int Cw = C / EPV;                       // <= 256 (checked by the host)
  const int RL = 256 / Cw;
  const int ck = threadIdx.x % Cw, rl = threadIdx.x / Cw;
  if (rl < RL) {
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll 4
    for (long r = r0 + rl; r < r1; r += RL) {
      float v[EPV];
      VecIO<T, EPV>::load(big + r * C + ck * EPV, v);
      const float tv = to_f32<T>(thin[r]);
#pragma unroll
      for (int i = 0; i < EPV; ++i) { s1[i] = fmaf(v[i], tv, s1[i]); s2[i] += big_is_dy ? v[i] : tv; }
    }
#pragma unroll
    for (int i = 0; i < EPV; ++i) {
      lds[((rl * 2 + 0) * Cw + ck) * EPV + i] = s1[i];
      lds[((rl * 2 + 1) * Cw + ck) * EPV + i] = s2[i];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, e = i % C;
    float acc = 0.f;
    for (int q = 0; q < RL; ++q) acc += lds[(q * 2 + which) * C + e];
    if (which == 0) dWp[(long)slot * C + e] = acc;
    else if (dbp && (big_is_dy || e == 0)) dbp[(long)slot * (big_is_dy ? C : 1) + (big_is_dy ? e : 0)] = acc;
  }
}

// out[i] = sum_s part[s][i]: 16 elements x 16 slot lanes per workgroup; lane j adds slots j, j+16, ... in order and
// the 16 lane sums are added in lane order -> a fixed summation tree, independent of the launch.
// one slot-lane's share of a [slots][n] -> [n] reduction: slots j, j+16, j+32, ...  Four independent accumulators keep four
// loads in flight per thread (a 1024-slot reduction was a 64-deep chain of dependent loads); every reduce kernel of the
// training path uses this function, so immediate, paired, batched and deferred reductions share one summation tree.
__device__ __forceinline__ float slot_lane_sum(const float* __restrict__ part, long n, long i, int j, int slots) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = j;
  for (; s + 48 < slots; s += 64) {
    a0 += part[(long)s * n + i];
    a1 += part[(long)(s + 16) * n + i];
    a2 += part[(long)(s + 32) * n + i];
    a3 += part[(long)(s + 48) * n + i];
  }
  for (; s < slots; s += 16) a0 += part[(long)s * n + i];
  return (a0 + a1) + (a2 + a3);
}

__global__ void __launch_bounds__(256)
reduce_slots_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int slots) {
  __shared__ float sm[16][17];
  const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + e;
  float a = 0.f;
  if (i < n)
    a = slot_lane_sum(part, n, i, j, slots);
  sm[j][e] = a;
  __syncthreads();
  if (j == 0 && i < n) {
    float t = sm[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += sm[q][e];
    out[i] = t;
  }
}

// the weight and bias partials of one gradient in ONE launch (same tree per element as reduce_slots_kernel)
__global__ void __launch_bounds__(256)
reduce_slots_pair_kernel(const float* __restrict__ partA, float* __restrict__ outA, long nA, const float* __restrict__ partB,
                         float* __restrict__ outB, long nB, int slots) {
  __shared__ float sm[16][17];
  const long blocksA = (nA + 15) / 16;
  const bool second = (long)blockIdx.x >= blocksA;
  const float* part = second ? partB : partA;
  float* out = second ? outB : outA;
  const long n = second ? nB : nA;
  const long b = second ? blockIdx.x - blocksA : blockIdx.x;
  const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
  const long i = b * 16 + e;
  float a = 0.f;
  if (i < n)
    a = slot_lane_sum(part, n, i, j, slots);
  sm[j][e] = a;
  __syncthreads();
  if (j == 0 && i < n) {
    float t = sm[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += sm[q][e];
    out[i] = t;
  }
}

static inline void reduce_slots_pair(const float* pA, float* oA, long nA, const float* pB, float* oB, long nB, int slots,
                                     hipStream_t s) {
  const long blocks = (nA + 15) / 16 + (oB ? (nB + 15) / 16 : 0);
  hipLaunchKernelGGL(reduce_slots_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pA, oA, nA, pB, oB, oB ? nB : 0, slots);
}

// several independent slot reductions in ONE launch (the gradients of one block's backward: dW3/db3, dW2/db2, dW1/db1,
// the norm sums, the residual conv): same per-element tree as reduce_slots_kernel, so deferring changes no bit
constexpr int REDUCE_MULTI_MAX = 12;
struct ReduceMulti {
  const float* part[REDUCE_MULTI_MAX];
  float* out[REDUCE_MULTI_MAX];
  long n[REDUCE_MULTI_MAX];
  int slots[REDUCE_MULTI_MAX];
  int out_t[REDUCE_MULTI_MAX];       // > 0: the n sums are a [n / out_t][out_t] matrix, written transposed ([out_t][n / out_t])
  int blk_end[REDUCE_MULTI_MAX];     // exclusive prefix end of each item's block range
  int count;
};
__global__ void __launch_bounds__(256)
reduce_slots_multi_kernel(ReduceMulti m) {
  __shared__ float sm[16][17];
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const long b = (long)blockIdx.x - (k ? m.blk_end[k - 1] : 0);
  const float* __restrict__ part = m.part[k];
  const long n = m.n[k];
  const int slots = m.slots[k];
  const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
  const long i = b * 16 + e;
  float a = 0.f;
  if (i < n)
    a = slot_lane_sum(part, n, i, j, slots);
  sm[j][e] = a;
  __syncthreads();
  if (j == 0 && i < n) {
    float t = sm[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += sm[q][e];
    const int ot = m.out_t[k];
    m.out[k][ot > 0 ? (i % ot) * (n / ot) + i / ot : i] = t;
  }
}

// batched form: blockIdx.y = sample; part [N][slots][n] -> out [N][n]
__global__ void __launch_bounds__(256)
reduce_slots_batched_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int slots) {
  __shared__ float sm[16][17];
  part += (long)blockIdx.y * slots * n;
  out += (long)blockIdx.y * n;
  const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + e;
  float a = 0.f;
  if (i < n)
    a = slot_lane_sum(part, n, i, j, slots);
  sm[j][e] = a;
  __syncthreads();
  if (j == 0 && i < n) {
    float t = sm[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += sm[q][e];
    out[i] = t;
  }
}

// ---- depthwise weight gradient:  dW[k][c] = sum_{n,o} G[n][o][c] * X[n][o*s - p + k][c],  db[c] = sum G ------------
// (conv: G = dL/dy on the output grid, X = input;  transposed conv: G = the layer INPUT, X = dL/dy, s = 2)
struct DwWg {
  int N, Dg, Hg, Wg, Dx, Hx, Wx, C, K, stride, pad;
  int lpv, vs, iters, slots;
  int kz_inner = 0;       // dw_wgrad_vec_kernel: kz as the fastest index of an XCD-aware 1-D grid
};

template <typename T, int VEC, int K>
__global__ void __launch_bounds__(256)
dw_wgrad_kernel(const T* __restrict__ g, const T* __restrict__ x, float* __restrict__ dWp, float* __restrict__ dbp,
                DwWg q) {
  extern __shared__ float lds[];   // [vs][C]
  const int n = blockIdx.y, slot = blockIdx.x;
  const int cv = threadIdx.x % q.lpv, vslot = threadIdx.x / q.lpv;
  const bool lane_ok = vslot < q.vs;
  const long vg = (long)q.Dg * q.Hg * q.Wg;
  const int C = q.C;
  const T* gn = g + (long)n * vg * C;
  const T* xn = x + (long)n * q.Dx * q.Hx * q.Wx * C;
  const long out_base = ((long)n * q.slots + slot);
  for (int kz = 0; kz < K; ++kz) {
    float acc[K * K][VEC];
    float bacc[VEC];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) bacc[i] = 0.f;
    for (int it = 0; it < q.iters; ++it) {
      long v = ((long)slot * q.iters + it) * q.vs + vslot;
      if (!lane_ok || v >= vg) continue;
      const int ox = (int)(v % q.Wg);
      long t = v / q.Wg;
      const int oy = (int)(t % q.Hg);
      const int oz = (int)(t / q.Hg);
      float gv[VEC];
      VecIO<T, VEC>::load(gn + v * C + cv * VEC, gv);
#pragma unroll
      for (int i = 0; i < VEC; ++i) bacc[i] += gv[i];
      const int iz = oz * q.stride - q.pad + kz;
      if (iz < 0 || iz >= q.Dx) continue;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * q.stride - q.pad + ky;
        if (iy < 0 || iy >= q.Hx) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int ix = ox * q.stride - q.pad + kx;
          if (ix < 0 || ix >= q.Wx) continue;
          float xv[VEC];
          VecIO<T, VEC>::load(xn + (((long)iz * q.Hx + iy) * q.Wx + ix) * C + cv * VEC, xv);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[ky * K + kx][i] = fmaf(gv[i], xv[i], acc[ky * K + kx][i]);
        }
      }
    }
    // reduce over the voxel slots of the workgroup (fixed order) and write this kz plane of taps
    for (int t = 0; t <= K * K; ++t) {     // t == K*K: the bias column (only once, with kz == 0)
      if (t == K * K && kz != 0) break;
      __syncthreads();
      if (lane_ok) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) lds[vslot * C + cv * VEC + i] = (t < K * K) ? acc[t < K * K ? t : 0][i] : bacc[i];
      }
      __syncthreads();
      for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f;
        for (int vv = 0; vv < q.vs; ++vv) a += lds[vv * C + c];
        if (t < K * K) dWp[(out_base * K * K * K + (long)kz * K * K + t) * C + c] = a;
        else if (dbp) dbp[out_base * C + c] = a;
      }
    }
  }
}

// 16-byte form for K = 3 at any stride (down blocks, transposed up blocks, volumes too small for the march):
// workgroup = (position slot, sample, kz); lane = (channel chunk of 16 bytes, position lane); 9 x EPV accumulators.
// Neighbouring taps / outputs share input lines through L1/L2; every parameter sum is reduced over the position lanes
// in lane order, then over slots by reduce_slots.
template <typename T>
__global__ void __launch_bounds__(256)
dw_wgrad_vec_kernel(const T* __restrict__ g, const T* __restrict__ x, float* __restrict__ dWp, float* __restrict__ dbp,
                    DwWg q, long rows_per_slot) {
  constexpr int EPV = 16 / (int)sizeof(T), K = 3;
  __shared__ float lds[5 * 256 * EPV];             // [value][position lane][C], five parameter rows per round
  // Round 6: the three kz workgroups of a (slot, sample) read the same G rows and (stride 2) share the odd X planes.  As gridDim.z they
  // were dispatched a whole grid apart; as the fastest index of an XCD-aware 1-D order they run back to back on ONE XCD and the second and
  // third find those lines in its L2 (q.kz_inner; the sums do not depend on the order).
  int slot = blockIdx.x, n = blockIdx.y, kz = blockIdx.z;
  if (q.kz_inner) {
    const int lb = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    kz = lb % 3;
    const int sn = lb / 3;
    slot = sn % q.slots;
    n = sn / q.slots;
  }
  const int C = q.C, Cw = C / EPV, PL = 256 / Cw;
  const int ck = threadIdx.x % Cw, pl = threadIdx.x / Cw;
  const long vg = (long)q.Dg * q.Hg * q.Wg;
  const T* gn = g + (long)n * vg * C + ck * EPV;
  const T* xn = x + (long)n * q.Dx * q.Hx * q.Wx * C + ck * EPV;
  float acc[K * K][EPV], bacc[EPV];
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int i = 0; i < EPV; ++i) bacc[i] = 0.f;
  const long v0 = (long)slot * rows_per_slot;
  const long v1 = v0 + rows_per_slot < vg ? v0 + rows_per_slot : vg;
  if (pl < PL) {
    for (long v = v0 + pl; v < v1; v += PL) {
      const int ox = (int)(v % q.Wg);
      const long t = v / q.Wg;
      const int oy = (int)(t % q.Hg), oz = (int)(t / q.Hg);
      float gv[EPV];
      VecIO<T, EPV>::load(gn + v * C, gv);
#pragma unroll
      for (int i = 0; i < EPV; ++i) bacc[i] += gv[i];
      const int iz = oz * q.stride - q.pad + kz;
      if (iz < 0 || iz >= q.Dx) continue;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * q.stride - q.pad + ky;
        if (iy < 0 || iy >= q.Hx) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int ix = ox * q.stride - q.pad + kx;
          if (ix < 0 || ix >= q.Wx) continue;
          float xv[EPV];
          VecIO<T, EPV>::load(xn + (((long)iz * q.Hx + iy) * q.Wx + ix) * C, xv);
#pragma unroll
          for (int i = 0; i < EPV; ++i) acc[ky * K + kx][i] = fmaf(gv[i], xv[i], acc[ky * K + kx][i]);
        }
      }
    }
  }
  const long out_base = (long)n * q.slots + slot;
  // the 9 tap sums + the bias column cross the position lanes through LDS five at a time (two rounds instead of ten: the tail
  // was a third of a 14^3 launch); per parameter the position lanes are added in lane order, as before
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (pl < PL) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int t = half * 5 + j;
#pragma unroll
        for (int i = 0; i < EPV; ++i) lds[(j * PL + pl) * C + ck * EPV + i] = t < K * K ? acc[t < K * K ? t : 0][i] : bacc[i];
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 5 * C; idx += 256) {
      const int j = idx / C, c = idx % C, t = half * 5 + j;
      if (t == K * K && (kz != 0 || !dbp)) continue;          // the bias column: kz == 0 only
      float a = 0.f;
      for (int vv = 0; vv < PL; ++vv) a += lds[(j * PL + vv) * C + c];
      if (t < K * K) dWp[(out_base * K * K * K + (long)kz * K * K + t) * C + c] = a;
      else dbp[out_base * C + c] = a;
    }
  }
}

// ---- GroupNorm(C groups = per (n,c)) backward ----------------------------------------------------------------------
// stats: s[n][slot][0][c] = sum dtn, s[..][1][c] = sum dtn * xhat,  xhat = (t - mean) * rstd
template <typename T>
__global__ void __launch_bounds__(256)
norm_bwd_stats_kernel(const T* __restrict__ dtn, const T* __restrict__ t, const float* __restrict__ mr,
                      float* __restrict__ stats, long rows, int C, int slots, long rows_per_slot) {
  extern __shared__ float lds[];
  const int n = blockIdx.y, slot = blockIdx.x;
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
  const T* dn = dtn + (long)n * rows * C;
  const T* tn = t + (long)n * rows * C;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int Cw = (C - c0) < 256 ? (C - c0) : 256;
    const int RL = 256 / Cw;
    const int c = threadIdx.x % Cw, rl = threadIdx.x / Cw;
    float s1 = 0.f, s2 = 0.f;
    if (rl < RL) {
      const float mean = mr[((long)n * 2 + 0) * C + c0 + c], rstd = mr[((long)n * 2 + 1) * C + c0 + c];
      for (long r = r0 + rl; r < r1; r += RL) {
        const float d = to_f32<T>(dn[r * C + c0 + c]);
        const float xh = (to_f32<T>(tn[r * C + c0 + c]) - mean) * rstd;
        s1 += d;
        s2 = fmaf(d, xh, s2);
      }
      lds[(rl * 2 + 0) * Cw + c] = s1;
      lds[(rl * 2 + 1) * Cw + c] = s2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cw; i += blockDim.x) {
      const int which = i / Cw, ch = i % Cw;
      float a = 0.f;
      for (int qq = 0; qq < RL; ++qq) a += lds[(qq * 2 + which) * Cw + ch];
      stats[(((long)n * slots + slot) * 2 + which) * C + c0 + ch] = a;
    }
    __syncthreads();
  }
}

// dt = rstd * gamma * (dtn - s1/V - xhat * s2/V)
template <typename T>
__global__ void __launch_bounds__(256)
norm_bwd_apply_kernel(const T* __restrict__ dtn, const T* __restrict__ t, const float* __restrict__ mr,
                      const float* __restrict__ gamma, const float* __restrict__ s, float inv_count,
                      T* __restrict__ dt, long rows, int C, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long n = i / (rows * C);
    const float mean = mr[(n * 2 + 0) * C + c], rstd = mr[(n * 2 + 1) * C + c];
    const float xh = (to_f32<T>(t[i]) - mean) * rstd;
    const float g = gamma ? gamma[c] : 1.f;
    const float v = rstd * g * (to_f32<T>(dtn[i]) - s[(n * 2 + 0) * C + c] * inv_count - xh * s[(n * 2 + 1) * C + c] * inv_count);
    dt[i] = from_f32<T>(v);
  }
}

// 16-byte form of the apply pass (C % (16/sizeof(T)) == 0): lane = (channel chunk, row lane), coefficients in registers
constexpr int NORM_APPLY_MAX_PART_C = 2048;      // channels when the statistics arrive in parts (LDS: 2 * C floats)
template <typename T>
__global__ void __launch_bounds__(256)
norm_bwd_apply_vec_kernel(const T* __restrict__ dtn, const T* __restrict__ t, const float* __restrict__ mr,
                          const float* __restrict__ gamma, const float* __restrict__ s, float inv_count,
                          T* __restrict__ dt, long rows, int C, long rows_per_slot, int crop_h = 0, int crop_w = 0,
                          int s_parts = 1) {
  // s: [s_parts][N][2][C], summed over the parts in order (1: the reduced sums of the statistics pass; > 1: the hidden-channel
  // chunks of norm_bwd_from_wgrad_kernel)
  // crop_h / crop_w > 0: the rows are a (D, crop_h, crop_w) grid whose FRONT faces (z, y or x == 0: the zero padding of an up
  // block's transposed conv) are dropped -- dt is the compact (D-1, crop_h-1, crop_w-1) grid the transposed conv's backward reads
  constexpr int EPV = 16 / (int)sizeof(T);
  const int n = blockIdx.y, slot = blockIdx.x;
  __shared__ float sm_s[2 * NORM_APPLY_MAX_PART_C];
  if (s_parts > 1) {      // the workgroup adds the parts once, together (a lane-private loop was 2 * EPV * parts scattered loads)
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
      float a = 0.f;
      for (int pp = 0; pp < s_parts; ++pp) a += s[((long)pp * gridDim.y + n) * 2 * C + i];
      sm_s[i] = a;
    }
    __syncthreads();
  }
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
  const long base = (long)n * rows * C;
  const long obase = crop_w > 0 ? (long)n * (rows / ((long)crop_h * crop_w) - 1) * (crop_h - 1) * (crop_w - 1) * C : base;
  const int chunks = C / EPV;
  for (int k0 = 0; k0 < chunks; k0 += 256) {
    const int Cw = (chunks - k0) < 256 ? (chunks - k0) : 256;
    const int RL = 256 / Cw;
    const int ck = threadIdx.x % Cw, rl = threadIdx.x / Cw;
    if (rl >= RL) continue;
    const int c = (k0 + ck) * EPV;
    float mean[EPV], rstd[EPV], rg[EPV], m1[EPV], m2[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) {
      mean[i] = mr[((long)n * 2 + 0) * C + c + i];
      rstd[i] = mr[((long)n * 2 + 1) * C + c + i];
      rg[i] = rstd[i] * (gamma ? gamma[c + i] : 1.f);
      m1[i] = (s_parts > 1 ? sm_s[c + i] : s[((long)n * 2 + 0) * C + c + i]) * inv_count;
      m2[i] = (s_parts > 1 ? sm_s[C + c + i] : s[((long)n * 2 + 1) * C + c + i]) * inv_count;
    }
#pragma unroll 2
    for (long r = r0 + rl; r < r1; r += RL) {
      float d[EPV], u[EPV], o[EPV];
      VecIO<T, EPV>::load(dtn + base + r * C + c, d);
      VecIO<T, EPV>::load(t + base + r * C + c, u);
#pragma unroll
      for (int i = 0; i < EPV; ++i) o[i] = rg[i] * (d[i] - m1[i] - (u[i] - mean[i]) * rstd[i] * m2[i]);
      long ro = r;
      if (crop_w > 0) {
        const int x = (int)(r % crop_w);
        const long q = r / crop_w;
        const int y = (int)(q % crop_h);
        const long z = q / crop_h;
        if (x == 0 || y == 0 || z == 0) continue;
        ro = ((z - 1) * (crop_h - 1) + (y - 1)) * (crop_w - 1) + (x - 1);
      }
      VecIO<T, EPV>::store(dt + obase + ro * C + c, o);
    }
  }
}

// ---- GroupNorm backward statistics from the per-sample weight-gradient partials -------------------------------------------
// With hp = W2 (gamma * xhat + beta) + b2 and dtn = W2^T dhp, the two sums the GroupNorm backward needs per (sample, channel),
//     S1[n][c] = sum_r dtn[r][c]            = sum_h W2[h][c] * q[n][h],      q[n][h]    = sum_r dhp[r][h]
//     S2[n][c] = sum_r dtn[r][c] xhat[r][c] = sum_h W2[h][c] * M[n][h][c],   M[n][h][c] = sum_r dhp[r][h] xhat[r][c]
// are contractions of the PER-SAMPLE weight-gradient sums M, q (the MFMA weight-gradient kernel run against xhat, split into bf16
// high and low parts, with per-sample slots) with the weights: no pass over the activations.  W2 is rounded to bf16 here, as the
// data-gradient GEMM's weight image is, so S1 / S2 are the sums of exactly the (unrounded) dtn that GEMM accumulates -- its
// epilogue (PYTC_RES_NORM_BWD) applies  dt = rstd*gamma * (dtn - S1/V - xhat * S2/V) = A*dtn + B*t + C  to those values, which
// makes dt orthogonal to (1, xhat) per channel as the two-pass form is.  (Statistics of the unrounded dtn combined with an apply
// pass over bf16(dtn) leave a component along xhat that the depthwise weight gradient amplifies: DESIGN.md section 4.4.)
// The parameter gradients follow from the same sums:
//     dW2[h][c] = sum_n gamma[c] * M[n][h][c] + beta[n][c] * q[n][h],   db2[h] = sum_n q[n][h]      (beta = b + mean * a of `ab`)
// workgroup = (16 channels, sample); thread = (channel, h lane): h = lane, lane + 16, ... over all hidden channels in chunks of
// 64 (q of a chunk is summed over the sample's slots first, four interleaved partial sums added in a fixed order).  It writes
// s_out [N][2][C], coef [N][3][C] = (A, B, C), the sample's term of dW2 IN PLACE of M[n], and q[n] -- term and q are then
// reduced over n like any other slot partial.
__global__ void __launch_bounds__(256)
norm_bwd_from_wgrad_kernel(float* __restrict__ M, const float* __restrict__ dbp, float* __restrict__ q,
                           const float* __restrict__ W2, const float* __restrict__ gamma, const float* __restrict__ ab,
                           const float* __restrict__ mr, float* __restrict__ s_out, float* __restrict__ coef, int N, int C,
                           int H, int sps, float inv_count) {
  __shared__ float sm_q[4][64];
  __shared__ float sm[2][16][17];
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 15, hl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool ok = c < C;
  const float g = ok ? (gamma ? gamma[c] : 1.f) : 0.f;
  const float mean = ok ? mr[((long)n * 2 + 0) * C + c] : 0.f, rstd = ok ? mr[((long)n * 2 + 1) * C + c] : 0.f;
  const float beta = ok ? fmaf(mean, ab[((long)n * 2 + 0) * C + c], ab[((long)n * 2 + 1) * C + c]) : 0.f;
  float* Mn = M + (long)n * H * C;
  float s1 = 0.f, s2 = 0.f;
  for (int h0 = 0; h0 < H; h0 += 64) {
    __syncthreads();                                  // the previous chunk's q is no longer read
    {
      const int hh = threadIdx.x & 63, part = threadIdx.x >> 6;
      float a = 0.f;
      if (h0 + hh < H) {
        // same order of additions as the plain loop, eight loads in flight (a level-0 sample has 256 slots: 64 dependent round trips
        // were 20 of this launch's 29 us)
        const float* __restrict__ src = dbp + (long)n * sps * H + h0 + hh;
        int j = part;
        for (; j + 28 < sps; j += 32) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = src[(long)(j + 4 * u) * H];
#pragma unroll
          for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; j < sps; j += 4) a += src[(long)j * H];
      }
      sm_q[part][hh] = a;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const float a = (sm_q[0][threadIdx.x] + sm_q[1][threadIdx.x]) + (sm_q[2][threadIdx.x] + sm_q[3][threadIdx.x]);
      sm_q[0][threadIdx.x] = a;
      if (blockIdx.x == 0 && h0 + (int)threadIdx.x < H) q[(long)n * H + h0 + threadIdx.x] = a;
    }
    __syncthreads();
    if (ok) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int hh = hl + 16 * k, h = h0 + hh;
        if (h >= H) break;
        const float w = to_f32<bf16_t>(from_f32<bf16_t>(W2[(long)h * C + c]));      // the data-gradient GEMM's weight
        const float m = Mn[(long)h * C + c], qq = sm_q[0][hh];
        s1 = fmaf(w, qq, s1);
        s2 = fmaf(w, m, s2);
        Mn[(long)h * C + c] = fmaf(g, m, beta * qq);
      }
    }
  }
  sm[0][hl][cl] = s1;
  sm[1][hl][cl] = s2;
  __syncthreads();
  if (hl == 0 && ok) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { a1 += sm[0][j][cl]; a2 += sm[1][j][cl]; }
    s_out[((long)n * 2 + 0) * C + c] = a1;
    s_out[((long)n * 2 + 1) * C + c] = a2;
    const float rg = rstd * g, m1 = a1 * inv_count, m2 = a2 * inv_count;
    const float B = -rg * rstd * m2;
    coef[((long)n * 3 + 0) * C + c] = rg;
    coef[((long)n * 3 + 1) * C + c] = B;
    coef[((long)n * 3 + 2) * C + c] = -rg * m1 - B * mean;
  }
}

// The same sums for H <= NBW_HMAX hidden channels with ONE pass over the q partials and no barrier inside the contraction loop (the
// chunked kernel above paid three barriers and one exposed load round trip per 64 hidden channels: 22 us at C = 256, H = 512).  Same
// order of additions everywhere (q: four interleaved slot partials, ((p0 + p1) + (p2 + p3)); s1 / s2: h ascending in steps of 16
// per lane, the 16 lanes in order) -> bit-identical to norm_bwd_from_wgrad_kernel.
constexpr int NBW_HMAX = 1024;
__global__ void __launch_bounds__(256)
norm_bwd_from_wgrad_flat_kernel(float* __restrict__ M, const float* __restrict__ dbp, float* __restrict__ q,
                                const float* __restrict__ W2, const float* __restrict__ gamma, const float* __restrict__ ab,
                                const float* __restrict__ mr, float* __restrict__ s_out, float* __restrict__ coef, int N, int C,
                                int H, int sps, float inv_count) {
  __shared__ float sm_q[4][NBW_HMAX];
  __shared__ float sm[2][16][17];
  const int n = blockIdx.y;
  const int cl = threadIdx.x & 15, hl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool ok = c < C;
  const float g = ok ? (gamma ? gamma[c] : 1.f) : 0.f;
  const float mean = ok ? mr[((long)n * 2 + 0) * C + c] : 0.f, rstd = ok ? mr[((long)n * 2 + 1) * C + c] : 0.f;
  const float beta = ok ? fmaf(mean, ab[((long)n * 2 + 0) * C + c], ab[((long)n * 2 + 1) * C + c]) : 0.f;
  float* Mn = M + (long)n * H * C;
  {
    const int hh = threadIdx.x & 63, part = threadIdx.x >> 6;
    for (int h = hh; h < H; h += 64) {
      const float* __restrict__ src = dbp + (long)n * sps * H + h;
      float a = 0.f;
      int j = part;
      for (; j + 28 < sps; j += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(j + 4 * u) * H];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
      }
      for (; j < sps; j += 4) a += src[(long)j * H];
      sm_q[part][h] = a;
    }
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += 256) {
    const float a = (sm_q[0][h] + sm_q[1][h]) + (sm_q[2][h] + sm_q[3][h]);
    sm_q[0][h] = a;
    if (blockIdx.x == 0) q[(long)n * H + h] = a;
  }
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  if (ok) {
    int h = hl;
    for (; h + 48 < H; h += 64) {
      float w[4], m[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        w[k] = W2[(long)(h + 16 * k) * C + c];
        m[k] = Mn[(long)(h + 16 * k) * C + c];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wb = to_f32<bf16_t>(from_f32<bf16_t>(w[k])), qq = sm_q[0][h + 16 * k];
        s1 = fmaf(wb, qq, s1);
        s2 = fmaf(wb, m[k], s2);
        Mn[(long)(h + 16 * k) * C + c] = fmaf(g, m[k], beta * qq);
      }
    }
    for (; h < H; h += 16) {
      const float wb = to_f32<bf16_t>(from_f32<bf16_t>(W2[(long)h * C + c]));
      const float m = Mn[(long)h * C + c], qq = sm_q[0][h];
      s1 = fmaf(wb, qq, s1);
      s2 = fmaf(wb, m, s2);
      Mn[(long)h * C + c] = fmaf(g, m, beta * qq);
    }
  }
  sm[0][hl][cl] = s1;
  sm[1][hl][cl] = s2;
  __syncthreads();
  if (hl == 0 && ok) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { a1 += sm[0][j][cl]; a2 += sm[1][j][cl]; }
    s_out[((long)n * 2 + 0) * C + c] = a1;
    s_out[((long)n * 2 + 1) * C + c] = a2;
    const float rg = rstd * g, m1 = a1 * inv_count, m2 = a2 * inv_count;
    const float B = -rg * rstd * m2;
    coef[((long)n * 3 + 0) * C + c] = rg;
    coef[((long)n * 3 + 1) * C + c] = B;
    coef[((long)n * 3 + 2) * C + c] = -rg * m1 - B * mean;
  }
}

static void launch_norm_bwd_from_wgrad(dim3 grid, hipStream_t s, float* M, const float* dbp, float* q, const float* W2, const float* gamma,
                                       const float* ab, const float* mr, float* s_out, float* coef, int N, int C, int H, int sps, float inv_count) {
  if (H <= NBW_HMAX && tuning_get("norm_bwd_from_wgrad_flat", 1))
    hipLaunchKernelGGL(norm_bwd_from_wgrad_flat_kernel, grid, dim3(256), 0, s, M, dbp, q, W2, gamma, ab, mr, s_out, coef, N, C, H, sps, inv_count);
  else
    hipLaunchKernelGGL(norm_bwd_from_wgrad_kernel, grid, dim3(256), 0, s, M, dbp, q, W2, gamma, ab, mr, s_out, coef, N, C, H, sps, inv_count);
}

// ---- strided depthwise backward-data (gather form): dx[i] = sum_k dy[(i + p - k)/s] * w[k] -------------------------
struct DwBd {
  int D, H, W, Do, Ho, Wo, C, K, stride, pad;
};
template <typename T>
__global__ void __launch_bounds__(256)
dwconv_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, DwBd g, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % g.C);
  long t = i / g.C;
  const int ix = (int)(t % g.W); t /= g.W;
  const int iy = (int)(t % g.H); t /= g.H;
  const int iz = (int)(t % g.D);
  const long n = t / g.D;
  const T* dn = dy + n * (long)g.Do * g.Ho * g.Wo * g.C;
  float acc = 0.f;
  for (int kz = 0; kz < g.K; ++kz) {
    const int tz = iz + g.pad - kz;
    if (tz < 0 || tz % g.stride || tz / g.stride >= g.Do) continue;
    for (int ky = 0; ky < g.K; ++ky) {
      const int ty = iy + g.pad - ky;
      if (ty < 0 || ty % g.stride || ty / g.stride >= g.Ho) continue;
      for (int kx = 0; kx < g.K; ++kx) {
        const int tx = ix + g.pad - kx;
        if (tx < 0 || tx % g.stride || tx / g.stride >= g.Wo) continue;
        acc = fmaf(to_f32<T>(dn[(((long)(tz / g.stride) * g.Ho + ty / g.stride) * g.Wo + tx / g.stride) * g.C + c]),
                   w[((long)(kz * g.K + ky) * g.K + kx) * g.C + c], acc);
      }
    }
  }
  dx[i] = from_f32<T>(acc);
}

// 16-byte form: lane = (input voxel, chunk of EPV channels)
template <typename T>
__global__ void __launch_bounds__(256)
dwconv_bwd_data_vec_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, DwBd g, long total,
                           const T* __restrict__ addend = nullptr) {
  // addend (nullable, shaped like dx): dx = conv^T(dy) + addend in fp32, one rounding -- the skip connection's gradient joins the down
  // block's data gradient here instead of in a separate three-pass add
  constexpr int EPV = 16 / (int)sizeof(T);
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Cw = g.C / EPV;
  const int c = (int)(i % Cw) * EPV;
  long t = i / Cw;
  const int ix = (int)(t % g.W); t /= g.W;
  const int iy = (int)(t % g.H); t /= g.H;
  const int iz = (int)(t % g.D);
  const long n = t / g.D;
  const T* dn = dy + n * (long)g.Do * g.Ho * g.Wo * g.C + c;
  float acc[EPV];
#pragma unroll
  for (int q = 0; q < EPV; ++q) acc[q] = 0.f;
  if (g.K == 3 && g.stride == 2) {
    // the MedNeXt down block: per axis an even input index meets tap 1 only (output i/2), an odd one taps 0 and 2 (outputs
    // (i+1)/2 and (i-1)/2) -- at most 8 of the 27 taps, found without the 27 x 3 runtime divisions of the general loop below
    // (363 -> ~100 us at 4 x 112^3 x 32).  Same tap order (ascending kz, ky, kx), so the sums are bit-identical.
    int kzs[2], ozs[2], kys[2], oys[2], kxs[2], oxs[2];
    auto taps = [](int i, int lim, int (&k)[2], int (&o)[2]) -> int {
      if ((i & 1) == 0) { k[0] = 1; o[0] = i >> 1; return o[0] < lim ? 1 : 0; }
      k[0] = 0; o[0] = (i + 1) >> 1; k[1] = 2; o[1] = (i - 1) >> 1;
      if (o[0] < lim) return 2;                       // o[1] < o[0]
      k[0] = 2; o[0] = o[1];
      return o[0] < lim ? 1 : 0;
    };
    const int nz = taps(iz, g.Do, kzs, ozs), ny = taps(iy, g.Ho, kys, oys), nx = taps(ix, g.Wo, kxs, oxs);
    for (int a = 0; a < nz; ++a)
      for (int b = 0; b < ny; ++b)
        for (int e = 0; e < nx; ++e) {
          float dv[EPV], wv[EPV];
          VecIO<T, EPV>::load(dn + (((long)ozs[a] * g.Ho + oys[b]) * g.Wo + oxs[e]) * g.C, dv);
          const float* wp = w + ((long)(kzs[a] * 3 + kys[b]) * 3 + kxs[e]) * g.C + c;
          VecIO<float, 4>::load(wp, *reinterpret_cast<float(*)[4]>(&wv[0]));
          if (EPV == 8) VecIO<float, 4>::load(wp + 4, *reinterpret_cast<float(*)[4]>(&wv[EPV == 8 ? 4 : 0]));
#pragma unroll
          for (int q = 0; q < EPV; ++q) acc[q] = fmaf(dv[q], wv[q], acc[q]);
        }
    if (addend) {
      float av[EPV];
      VecIO<T, EPV>::load(addend + i * EPV, av);
#pragma unroll
      for (int q = 0; q < EPV; ++q) acc[q] += av[q];
    }
    VecIO<T, EPV>::store(dx + i * EPV, acc);
    return;
  }
  for (int kz = 0; kz < g.K; ++kz) {
    const int tz = iz + g.pad - kz;
    if (tz < 0 || tz % g.stride || tz / g.stride >= g.Do) continue;
    for (int ky = 0; ky < g.K; ++ky) {
      const int ty = iy + g.pad - ky;
      if (ty < 0 || ty % g.stride || ty / g.stride >= g.Ho) continue;
      for (int kx = 0; kx < g.K; ++kx) {
        const int tx = ix + g.pad - kx;
        if (tx < 0 || tx % g.stride || tx / g.stride >= g.Wo) continue;
        float dv[EPV], wv[EPV];
        VecIO<T, EPV>::load(dn + (((long)(tz / g.stride) * g.Ho + ty / g.stride) * g.Wo + tx / g.stride) * g.C, dv);
        const float* wp = w + ((long)(kz * g.K + ky) * g.K + kx) * g.C + c;
        VecIO<float, 4>::load(wp, *reinterpret_cast<float(*)[4]>(&wv[0]));
        if (EPV == 8) VecIO<float, 4>::load(wp + 4, *reinterpret_cast<float(*)[4]>(&wv[EPV == 8 ? 4 : 0]));
#pragma unroll
        for (int q = 0; q < EPV; ++q) acc[q] = fmaf(dv[q], wv[q], acc[q]);
      }
    }
  }
  if (addend) {
    float av[EPV];
    VecIO<T, EPV>::load(addend + i * EPV, av);
#pragma unroll
    for (int q = 0; q < EPV; ++q) acc[q] += av[q];
  }
  VecIO<T, EPV>::store(dx + i * EPV, acc);
}

static int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 16384 ? b : 16384); }

}  // namespace pytc

using namespace pytc;

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32, name)                 \
  if ((dtype) == PYTC_BF16) { CALL_BF16; }                            \
  else if ((dtype) == PYTC_F32) { CALL_F32; }                         \
  else { PYTC_REQUIRE(false, name ": bad dtype"); }

extern "C" int pytc_gelu(const void* x, const void* dy, void* out, int64_t n, int dtype, void* stream) {
  PYTC_REQUIRE(x && out && n > 0, "gelu: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gelu_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, (long)n),
             hipLaunchKernelGGL(gelu_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)out, (long)n),
             "gelu")
  PYTC_LAUNCH_CHECK("gelu");
  return PYTC_OK;
}

extern "C" int pytc_add_inplace(void* y, const void* x, int64_t n, int dtype, void* stream) {
  PYTC_REQUIRE(x && y && n > 0, "add_inplace: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)y, (const bf16_t*)x, (long)n),
             hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (float*)y, (const float*)x, (long)n),
             "add_inplace")
  PYTC_LAUNCH_CHECK("add_inplace");
  return PYTC_OK;
}

extern "C" int pytc_copy_zero_front(const void* src, void* dst, int N, const int32_t* dims, int C, int dtype, void* stream) {
  PYTC_REQUIRE(src && dst && dims && N >= 1 && dims[0] >= 1 && dims[1] >= 1 && dims[2] >= 1 && C >= 1, "copy_zero_front: bad arguments");
  const int esz = dtype == PYTC_BF16 ? 2 : 4;
  PYTC_REQUIRE((C * esz) % 16 == 0, "copy_zero_front: a voxel's channels must be a multiple of 16 bytes (C = %d)", C);
  const int ppv = C * esz / 16;
  const long pieces = (long)N * dims[0] * dims[1] * dims[2] * ppv;
  const long blocks = (pieces + 255) / 256;
  hipLaunchKernelGGL(copy_zero_front_kernel, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, pieces, ppv,
                     dims[0], dims[1], dims[2]);
  PYTC_LAUNCH_CHECK("copy_zero_front");
  return PYTC_OK;
}

extern "C" int pytc_pw_wgrad_slots(int64_t rows_total) {
  // the workspace bound: >= every slot count a launch may pick (wgrad_mfma_slots: down to 256 rows per slot on small problems)
  long s = rows_total / 256;
  return (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
}

static int wg_tile16(int C) { return C % 64 == 0 ? 4 : (C % 32 == 0 ? 2 : (C % 16 == 0 ? 1 : 0)); }

// row slots of the MFMA weight-gradient launch: pytc_pw_wgrad_slots() is the workspace bound; the launch uses fewer row slots when
// the channel tiles already supply workgroups
static int wgrad_mfma_slots(long rows_total, int C_in, int C_out, int slots) {
  const int mt = wg_tile16(C_out), nt = wg_tile16(C_in);
  const long tiles = (long)(C_out / (16 * mt)) * (C_in / (16 * nt));
  long want = rows_total / 2048;
  const long cap = 1024 / tiles > 1 ? 1024 / tiles : 1;
  want = want < 1 ? 1 : (want > cap ? cap : want);
  // a launch that needs the cap would run 1024 workgroups over 256 CUs x (2 | 3 | 4) resident ones: 1.33 rounds, the last a
  // third full.  Cut it to whole rounds of resident workgroups (the slot count only groups rows: sums stay in slot order)
  const int per_cu = mt * nt >= 8 ? 2 : (mt * nt >= 4 ? 3 : 4);
  const long resident = 256L * per_cu / tiles;
  if (tuning_get("wgrad_whole_rounds", 1) && resident >= 8 && want > resident) want = (want / resident) * resident;
  // deep levels (14^3 / 7^3 voxels per sample): 2048-row slots leave 160 / 128 workgroups, each a chain of 11-17 dependent
  // row blocks per wave (42-48 us for 17 / 4 MB of operands).  Up to ~2 workgroups per CU, slots shrink to >= 256 rows
  // (2 row blocks per wave: what the two-deep prefetch needs); the extra partials are a few MB
  if (tuning_get("wgrad_small_split", 1) && want * tiles < 512) {
    long more = 512 / tiles, most = rows_total / 256;
    more = more < most ? more : most;
    if (more > want) want = more;
  }
  return want < slots ? (int)want : slots;
}

template <int MT>
static void launch_wgrad_mfma(int nt, dim3 grid, hipStream_t s, const bf16_t* x, const float* ab, const bf16_t* dy,
                              float* dWp, float* dbp, long rows_total, long rps_sample, int C_in, int C_out, long rps, int x_act,
                              int sps = 0, int ab_mode = 0) {
  switch (nt) {
    case 4: hipLaunchKernelGGL((pw_wgrad_mfma_kernel<MT, 4>), grid, dim3(256), 0, s, x, ab, dy, dWp, dbp, rows_total, rps_sample, C_in, C_out, rps, x_act, sps, ab_mode); break;
    case 2: hipLaunchKernelGGL((pw_wgrad_mfma_kernel<MT, 2>), grid, dim3(256), 0, s, x, ab, dy, dWp, dbp, rows_total, rps_sample, C_in, C_out, rps, x_act, sps, ab_mode); break;
    default: hipLaunchKernelGGL((pw_wgrad_mfma_kernel<MT, 1>), grid, dim3(256), 0, s, x, ab, dy, dWp, dbp, rows_total, rps_sample, C_in, C_out, rps, x_act, sps, ab_mode); break;
  }
}

// reduce = false: everything but the final slot reduction (`db` then only says whether bias partials are wanted);
// *slots_out = partial slots written (dW partials at workspace[0 .. slots*C_out*C_in), db partials right after)
static int pw_wgrad_impl(const void* x, const float* ab, const void* dy, float* dW, float* db, float* workspace,
                         int N, int64_t rows_per_sample, int C_in, int C_out, int dtype, int x_act, void* stream,
                         bool reduce, int* slots_out) {
  PYTC_REQUIRE(x && dy && (dW || !reduce) && workspace && N >= 1 && rows_per_sample >= 1, "pw_wgrad: bad arguments");
  PYTC_REQUIRE(x_act == PYTC_ACT_NONE || x_act == PYTC_ACT_GELU, "pw_wgrad: bad x_act");
  const long rows_total = (long)N * rows_per_sample;
  const int mt = wg_tile16(C_out), nt = wg_tile16(C_in);
  // every workgroup ends with a cross-wave reduction and a C_out x C_in partial, so short slots (8 row blocks at level 1) spent
  // most of their time there: >= 2048 rows per slot, ~1024 workgroups at most (wgrad_mfma_slots)
  int slots = pytc_pw_wgrad_slots(rows_total);
  if (dtype == PYTC_BF16 && mt && nt) slots = wgrad_mfma_slots(rows_total, C_in, C_out, slots);
  const long rps = (rows_total + slots - 1) / slots;
  float* dWp = workspace;
  float* dbp = workspace + (long)slots * C_out * C_in;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PYTC_BF16 && mt && nt && tuning_get("wgrad_valu", 0) == 0) {
    // bf16, channel counts in multiples of 16: MFMA path (transpose reads from a wave-private LDS image)
    dim3 grid(slots, (C_out / (16 * mt)) * (C_in / (16 * nt)));
    const bf16_t* xp = (const bf16_t*)x;
    const bf16_t* dp = (const bf16_t*)dy;
    float* dbq = db ? dbp : nullptr;
    if (mt == 4) launch_wgrad_mfma<4>(nt, grid, s, xp, ab, dp, dWp, dbq, rows_total, (long)rows_per_sample, C_in, C_out, rps, x_act);
    else if (mt == 2) launch_wgrad_mfma<2>(nt, grid, s, xp, ab, dp, dWp, dbq, rows_total, (long)rows_per_sample, C_in, C_out, rps, x_act);
    else launch_wgrad_mfma<1>(nt, grid, s, xp, ab, dp, dWp, dbq, rows_total, (long)rows_per_sample, C_in, C_out, rps, x_act);
  } else if (!ab && x_act == PYTC_ACT_NONE && (C_in == 1 || C_out == 1) && (C_in * C_out) % (dtype == PYTC_BF16 ? 8 : 4) == 0 &&
             C_in * C_out <= (dtype == PYTC_BF16 ? 2048 : 1024)) {
    const int Cb = C_in * C_out, big_is_dy = C_in == 1;
    const void* big = big_is_dy ? dy : x;
    const void* thin = big_is_dy ? x : dy;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(pw_wgrad_thin_kernel<bf16_t>, dim3(slots), dim3(256), 0, s, (const bf16_t*)big, (const bf16_t*)thin, dWp, db ? dbp : nullptr, rows_total, Cb, rps, big_is_dy),
               hipLaunchKernelGGL(pw_wgrad_thin_kernel<float>, dim3(slots), dim3(256), 0, s, (const float*)big, (const float*)thin, dWp, db ? dbp : nullptr, rows_total, Cb, rps, big_is_dy),
               "pw_wgrad")
  } else {
    dim3 grid(slots, ((C_out + WG_TO - 1) / WG_TO) * ((C_in + WG_TK - 1) / WG_TK)), block(256);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(pw_wgrad_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, ab, (const bf16_t*)dy, dWp, db ? dbp : nullptr, rows_total, (long)rows_per_sample, C_in, C_out, rps, slots, x_act),
               hipLaunchKernelGGL(pw_wgrad_kernel<float>, grid, block, 0, s, (const float*)x, ab, (const float*)dy, dWp, db ? dbp : nullptr, rows_total, (long)rows_per_sample, C_in, C_out, rps, slots, x_act),
               "pw_wgrad")
  }
  const long nW = (long)C_out * C_in;
  if (slots_out) *slots_out = slots;
  if (reduce) reduce_slots_pair(dWp, dW, nW, dbp, db, (long)C_out, slots, s);
  PYTC_LAUNCH_CHECK("pw_wgrad");
  return PYTC_OK;
}

extern "C" int pytc_pw_wgrad(const void* x, const float* ab, const void* dy, float* dW, float* db, float* workspace,
                             int N, int64_t rows_per_sample, int C_in, int C_out, int dtype, int x_act, void* stream) {
  return pw_wgrad_impl(x, ab, dy, dW, db, workspace, N, rows_per_sample, C_in, C_out, dtype, x_act, stream, true, nullptr);
}

extern "C" int pytc_pw_wgrad_partial(const void* x, const float* ab, const void* dy, float* workspace, int want_db, int N,
                                     int64_t rows_per_sample, int C_in, int C_out, int dtype, int x_act, int* slots_out,
                                     void* stream) {
  PYTC_REQUIRE(slots_out, "pw_wgrad_partial: null slots_out");
  float* db_flag = want_db ? workspace : nullptr;      // non-null = "write bias partials" (never dereferenced as output)
  return pw_wgrad_impl(x, ab, dy, nullptr, db_flag, workspace, N, rows_per_sample, C_in, C_out, dtype, x_act, stream, false,
                       slots_out);
}

extern "C" int pytc_pw_wgrad_dgrad_supported(int C_in, int C_out, int dtype) {
  return (dtype == PYTC_BF16 && C_out == 32 && (C_in == 64 || C_in == 32 || (C_in == 128 && tuning_get("wgrad_dgrad_wide", 1) != 0)) &&
          tuning_get("wgrad_dgrad_fused", 1) != 0) ? 1 : 0;
}

extern "C" int pytc_pw_wgrad_dgrad_partial(const void* x, const void* dy, const void* w_t_paired, void* dx, float* workspace,
                                           int want_db, int N, int64_t rows_per_sample, int C_in, int C_out, int dtype,
                                           int* slots_out, void* stream) {
  PYTC_REQUIRE(x && dy && w_t_paired && dx && workspace && slots_out && N >= 1 && rows_per_sample >= 1, "pw_wgrad_dgrad: bad arguments");
  PYTC_REQUIRE(pytc_pw_wgrad_dgrad_supported(C_in, C_out, dtype), "pw_wgrad_dgrad: bf16, C_out = 32, C_in in {32, 64, 128} (C_in=%d C_out=%d)", C_in, C_out);
  const long rows_total = (long)N * rows_per_sample;
  int slots = wgrad_mfma_slots(rows_total, C_in, C_out, pytc_pw_wgrad_slots(rows_total));
  // one workgroup per row slot holds ALL channels here (the plain launch splits 32 x 64 into channel tiles only above 64 x 64: same grid)
  const long rps = (rows_total + slots - 1) / slots;
  float* dWp = workspace;
  float* dbp = workspace + (long)slots * C_out * C_in;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(slots, 1);
  const bf16_t* xp = (const bf16_t*)x;
  const bf16_t* dp = (const bf16_t*)dy;
  if (C_in == 128)     // wide hidden layer (the 64 -> 128 -> 32 up block): pw_wgrad_dgrad_wide_kernel; g' = the derivative of gelu_fast
    hipLaunchKernelGGL((pw_wgrad_dgrad_wide_kernel<8>), grid, dim3(256), 0, s, xp, dp, (const bf16x8_t*)w_t_paired, dWp, want_db ? dbp : nullptr,
                       (bf16_t*)dx, rows_total, rps);
  else if (C_in == 64)
    hipLaunchKernelGGL((pw_wgrad_mfma_kernel<2, 4, true>), grid, dim3(256), 0, s, xp, (const float*)nullptr, dp, dWp, want_db ? dbp : nullptr, rows_total,
                       (long)rows_per_sample, C_in, C_out, rps, PYTC_ACT_GELU, 0, 0, (const bf16x8_t*)w_t_paired, (bf16_t*)dx);
  else
    hipLaunchKernelGGL((pw_wgrad_mfma_kernel<2, 2, true>), grid, dim3(256), 0, s, xp, (const float*)nullptr, dp, dWp, want_db ? dbp : nullptr, rows_total,
                       (long)rows_per_sample, C_in, C_out, rps, PYTC_ACT_GELU, 0, 0, (const bf16x8_t*)w_t_paired, (bf16_t*)dx);
  *slots_out = slots;
  PYTC_LAUNCH_CHECK("pw_wgrad_dgrad");
  return PYTC_OK;
}

extern "C" int pytc_reduce_slots_multi(const pytc_reduce_item* items, int n_items, void* stream) {
  PYTC_REQUIRE(items && n_items >= 1 && n_items <= REDUCE_MULTI_MAX, "reduce_slots_multi: 1..%d items", REDUCE_MULTI_MAX);
  ReduceMulti m;
  int blocks = 0;
  for (int k = 0; k < n_items; ++k) {
    PYTC_REQUIRE(items[k].part && items[k].out && items[k].n >= 1 && items[k].slots >= 1, "reduce_slots_multi: bad item %d", k);
    PYTC_REQUIRE(items[k].out_t >= 0 && (items[k].out_t == 0 || items[k].n % items[k].out_t == 0), "reduce_slots_multi: item %d: out_t must divide n", k);
    m.part[k] = items[k].part; m.out[k] = items[k].out; m.n[k] = (long)items[k].n; m.slots[k] = items[k].slots;
    m.out_t[k] = items[k].out_t;
    blocks += (int)((items[k].n + 15) / 16);
    m.blk_end[k] = blocks;
  }
  m.count = n_items;
  hipLaunchKernelGGL(reduce_slots_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m);
  PYTC_LAUNCH_CHECK("reduce_slots_multi");
  return PYTC_OK;
}

/* GroupNorm-fed expand conv: weight-gradient sums AND the norm-backward statistics from one pass over (t, dhp) (see
   norm_bwd_from_wgrad_kernel).  bf16, C and C_hid multiples of 16.  workspace (pytc_pw_wgrad_groupnorm_ws_elems floats):
     [N*sps][C_hid*C] dW partials | [N*sps][C_hid] bias partials | term [N][C_hid*C] | q [N][C_hid]
   with sps = pytc_pw_wgrad_groupnorm_sps (sps == 1: the samples' terms are left IN the dW partials region, `term` stays unused).
   On return  dW2 = sum_n term[n],  db2 = sum_n q[n]  (left to the caller's slot reduction, N slots each), s_out [N][2][C] = (sum dtn, sum dtn * xhat) and coef [N][3][C] = (A, B, C) for the data-gradient
   GEMM's PYTC_RES_NORM_BWD epilogue (count = voxels in the statistics). */
extern "C" int pytc_pw_wgrad_groupnorm_sps(int N, int64_t rows_per_sample, int C, int C_hid) {
  const long rows_total = (long)N * rows_per_sample;
  const int slots = wgrad_mfma_slots(rows_total, C, C_hid, pytc_pw_wgrad_slots(rows_total));
  const int sps = slots / N;
  return sps < 1 ? 1 : sps;
}

extern "C" int pytc_pw_wgrad_groupnorm_supported(int C, int C_hid, int dtype) {
  return dtype == PYTC_BF16 && wg_tile16(C) != 0 && wg_tile16(C_hid) != 0;
}

extern "C" int64_t pytc_pw_wgrad_groupnorm_ws_elems(int N, int64_t rows_per_sample, int C, int C_hid) {
  const long per = (long)C_hid * C + C_hid;
  return (int64_t)N * pytc_pw_wgrad_groupnorm_sps(N, rows_per_sample, C, C_hid) * per + (int64_t)N * per;
}

extern "C" int pytc_pw_wgrad_groupnorm(const void* t, const float* mean_rstd, const float* ab, const void* dhp, const float* W2,
                                       const float* gamma, float count, float* s_out, float* coef, float* workspace, int N,
                                       int64_t rows_per_sample, int C, int C_hid, int dtype, void* stream) {
  PYTC_REQUIRE(t && mean_rstd && ab && dhp && W2 && s_out && coef && workspace && N >= 1 && N <= 65535 && rows_per_sample >= 1 &&
               count > 0.f, "pw_wgrad_groupnorm: bad arguments");
  PYTC_REQUIRE(pytc_pw_wgrad_groupnorm_supported(C, C_hid, dtype), "pw_wgrad_groupnorm: bf16 with C, C_hid multiples of 16 (got %d, %d)", C, C_hid);
  const int sps = pytc_pw_wgrad_groupnorm_sps(N, rows_per_sample, C, C_hid);
  const long rps = (rows_per_sample + sps - 1) / sps;
  const long nW = (long)C_hid * C;
  float* dWp = workspace;                                  // [N][sps][C_hid][C]
  float* dbp = dWp + (long)N * sps * nW;                   // [N][sps][C_hid]
  float* term = dbp + (long)N * sps * C_hid;               // M [N][C_hid][C] -> the samples' terms of dW2
  float* qv = term + (long)N * nW;                         // q [N][C_hid]
  hipStream_t s = (hipStream_t)stream;
  const int mt = wg_tile16(C_hid), nt = wg_tile16(C);
  dim3 grid(N * sps, (C_hid / (16 * mt)) * (C / (16 * nt)));
  const bf16_t* xp = (const bf16_t*)t;
  const bf16_t* dp = (const bf16_t*)dhp;
  const long rows_total = (long)N * rows_per_sample;
  if (mt == 4) launch_wgrad_mfma<4>(nt, grid, s, xp, mean_rstd, dp, dWp, dbp, rows_total, (long)rows_per_sample, C, C_hid, rps, PYTC_ACT_NONE, sps, 1);
  else if (mt == 2) launch_wgrad_mfma<2>(nt, grid, s, xp, mean_rstd, dp, dWp, dbp, rows_total, (long)rows_per_sample, C, C_hid, rps, PYTC_ACT_NONE, sps, 1);
  else launch_wgrad_mfma<1>(nt, grid, s, xp, mean_rstd, dp, dWp, dbp, rows_total, (long)rows_per_sample, C, C_hid, rps, PYTC_ACT_NONE, sps, 1);
  // one slot per sample (the 14^3 / 7^3 levels): the partials ARE the per-sample sums -- the epilogue kernel works on them in place and the
  // samples' terms of dW2 stay in the partials region (no reduction launch: 12 us + a launch boundary per deep block)
  float* M = sps == 1 ? dWp : term;
  if (sps > 1) hipLaunchKernelGGL(reduce_slots_batched_kernel, dim3(ceil_div(nW, 16), N), dim3(256), 0, s, dWp, term, nW, sps);
  launch_norm_bwd_from_wgrad(dim3((C + 15) / 16, N), s, M, dbp, qv, W2, gamma, ab, mean_rstd, s_out, coef, N, C, C_hid, sps, 1.0f / count);
  PYTC_LAUNCH_CHECK("pw_wgrad_groupnorm");
  return PYTC_OK;
}

/* Level-0 mixer backward with the hidden pre-activation rebuilt (mixer_bwd_rc_kernel): from t, dy and the block's weights to
   dhp = (W3^T dy) * gelu'(hp), the weight-gradient partials of the projecting conv and -- gn != 0 -- everything pytc_pw_wgrad_groupnorm
   returns.  workspace (pytc_mixer_bwd_rc_ws_elems floats), S = N * sps slots:
     dW3 partials [S][32][C_hid] | db3 partials [S][32] | gn: M partials [S][C_hid][32] | q partials [S][C_hid] | term [N][C_hid][32] | q [N][C_hid] */
extern "C" int pytc_mixer_bwd_rc_supported(int C, int C_hid, int C_out, int dtype) {      // 0 no, 1 without the GroupNorm form, 2 both
  if (!(dtype == PYTC_BF16 && C == 32 && C_out == 32 && (C_hid == 64 || C_hid == 32 || C_hid == 96)) || tuning_get("mixer_bwd_rc", 1) == 0) return 0;
  return (C_hid == 96 || tuning_get("mixer_bwd_rc_gn", 1) == 0) ? 1 : 2;          // 96: the GroupNorm form's accumulators do not fit 256 registers (68 spilled)
}

extern "C" int pytc_mixer_bwd_rc_sps(int N, int64_t rows_per_sample, int C_hid) {
  const long rows_total = (long)N * rows_per_sample;
  int slots = wgrad_mfma_slots(rows_total, C_hid, 32, pytc_pw_wgrad_slots(rows_total));
  // knob (measurement): fewer, longer slots -- halves the partials the block's reduction launch reads
  const int div = tuning_get("mixer_bwd_rc_slot_div", 1);
  if (div > 1 && slots / div >= 512) slots /= div;
  const int sps = slots / N;
  return sps < 1 ? 1 : sps;
}

extern "C" int64_t pytc_mixer_bwd_rc_ws_elems(int N, int64_t rows_per_sample, int C_hid, int gn) {
  const long S = (long)N * pytc_mixer_bwd_rc_sps(N, rows_per_sample, C_hid);
  const long per = 32L * C_hid;
  return (int64_t)(S * (per + 32) + (gn ? S * (per + C_hid) + (long)N * (per + C_hid) : 0));
}

template <int HT>
static void launch_mixer_bwd_rc(const MixBwd& q, int gn, int slots, hipStream_t s) {
  if (gn) hipLaunchKernelGGL((mixer_bwd_rc_kernel<HT, true>), dim3(slots), dim3(256), 0, s, q);
  else hipLaunchKernelGGL((mixer_bwd_rc_kernel<HT, false>), dim3(slots), dim3(256), 0, s, q);
}

extern "C" int pytc_mixer_bwd_rc(const void* t, const float* ab, const float* mean_rstd, const void* dy, const void* w2_paired,
                                 const float* b2, const void* w3t_paired, const float* W2, const float* gamma, float count, void* dhp,
                                 float* workspace, float* s_out, float* coef, int want_db3, int gn, int N, int64_t rows_per_sample,
                                 int C, int C_hid, int C_out, int dtype, int* slots_out, void* stream) {
  PYTC_REQUIRE(t && ab && dy && w2_paired && b2 && w3t_paired && dhp && workspace && slots_out && N >= 1 && N <= 65535 && rows_per_sample >= 1,
               "mixer_bwd_rc: bad arguments");
  PYTC_REQUIRE(pytc_mixer_bwd_rc_supported(C, C_hid, C_out, dtype) >= (gn ? 2 : 1),
               "mixer_bwd_rc: bf16, C = C_out = 32, C_hid in {32, 64, 96} (96: gn = 0 only) (got %d, %d, %d, gn %d)", C, C_hid, C_out, gn);
  PYTC_REQUIRE(!gn || (mean_rstd && W2 && s_out && coef && count > 0.f), "mixer_bwd_rc: the GroupNorm form needs mean_rstd, W2, s_out, coef, count");
  const int sps = pytc_mixer_bwd_rc_sps(N, rows_per_sample, C_hid);
  const long S = (long)N * sps, per = 32L * C_hid;
  MixBwd q{};
  q.t = (const bf16_t*)t; q.ab = ab; q.mr = mean_rstd; q.dy = (const bf16_t*)dy;
  q.w2p = (const bf16x8_t*)w2_paired; q.b2 = b2; q.w3t = (const bf16x8_t*)w3t_paired;
  q.dW3p = workspace; q.db3p = q.dW3p + S * per;
  q.dW2p = q.db3p + S * 32; q.db2p = q.dW2p + S * per;
  float* term = q.db2p + S * C_hid;
  float* qv = term + (long)N * per;
  q.dhp = (bf16_t*)dhp;
  q.rows_per_sample = rows_per_sample; q.rows_per_slot = (rows_per_sample + sps - 1) / sps;
  q.sps = sps; q.want_db3 = want_db3; q.probe = tuning_get("mixer_bwd_rc_probe", 0);
  hipStream_t s = (hipStream_t)stream;
  if (C_hid == 64) launch_mixer_bwd_rc<4>(q, gn, (int)S, s);
  else if (C_hid == 96) hipLaunchKernelGGL((mixer_bwd_rc_kernel<6, false>), dim3((unsigned)S), dim3(256), 0, s, q);
  else launch_mixer_bwd_rc<2>(q, gn, (int)S, s);
  if (gn) {
    hipLaunchKernelGGL(reduce_slots_batched_kernel, dim3(ceil_div(per, 16), N), dim3(256), 0, s, q.dW2p, term, per, sps);
    launch_norm_bwd_from_wgrad(dim3(2, N), s, term, q.db2p, qv, W2, gamma, ab, mean_rstd, s_out, coef, N, 32, C_hid, sps, 1.0f / count);
  }
  *slots_out = (int)S;
  PYTC_LAUNCH_CHECK("mixer_bwd_rc");
  return PYTC_OK;
}

/* the apply pass of pytc_norm_bwd with the statistics given: s_in [s_parts][N][2][C], added over the parts (1 for the output
   of pytc_norm_bwd_stats).  crop_grid (nullable, int32[3] = the (D, H, W) grid of the rows): drop the front faces, dt is the
   compact (D-1, H-1, W-1) grid. */
extern "C" int pytc_norm_bwd_apply(const void* dtn, const void* t, const float* mean_rstd, const float* gamma, const float* s_in,
                                   int s_parts, void* dt, int N, int64_t rows, int C, float count, int dtype,
                                   const int32_t* crop_grid, void* stream) {
  PYTC_REQUIRE(dtn && t && mean_rstd && s_in && s_parts >= 1 && dt && count > 0.f, "norm_bwd_apply: bad arguments");
  const bool vec = dtype == PYTC_BF16 ? (C % 8 == 0) : (C % 4 == 0);
  PYTC_REQUIRE(vec, "norm_bwd_apply: C must be a multiple of %d", dtype == PYTC_BF16 ? 8 : 4);
  PYTC_REQUIRE(s_parts == 1 || C <= NORM_APPLY_MAX_PART_C, "norm_bwd_apply: statistics in parts need C <= %d", NORM_APPLY_MAX_PART_C);
  int ch = 0, cw = 0;
  if (crop_grid) {
    PYTC_REQUIRE((long)crop_grid[0] * crop_grid[1] * crop_grid[2] == rows && crop_grid[0] > 1 && crop_grid[1] > 1 && crop_grid[2] > 1,
                 "norm_bwd_apply: crop grid does not match the rows");
    ch = crop_grid[1]; cw = crop_grid[2];
  }
  const int slots = colstats_slots(rows);
  const long rps = (rows + slots - 1) / slots;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(slots, N), block(256);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(norm_bwd_apply_vec_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)dtn, (const bf16_t*)t, mean_rstd, gamma, s_in, 1.0f / count, (bf16_t*)dt, (long)rows, C, rps, ch, cw, s_parts),
             hipLaunchKernelGGL(norm_bwd_apply_vec_kernel<float>, grid, block, 0, s, (const float*)dtn, (const float*)t, mean_rstd, gamma, s_in, 1.0f / count, (float*)dt, (long)rows, C, rps, ch, cw, s_parts),
             "norm_bwd_apply")
  PYTC_LAUNCH_CHECK("norm_bwd_apply");
  return PYTC_OK;
}

// z-march form of the stride-1 3x3x3 case (dwconv_kernels.hip)
namespace pytc {
int dw_wgrad_march_slots(int N, int D, int H, int W, int C, int K, int stride, int dtype);
void dw_wgrad_march_launch(const void* gr, const void* x, float* dWp, float* dbp, int N, int D, int H, int W, int C,
                           int dtype, hipStream_t s);
}
static int march_slots(int N, const int32_t* gd, const int32_t* xd, int C, int K, int stride, int dtype) {
  if (gd[0] != xd[0] || gd[1] != xd[1] || gd[2] != xd[2]) return 0;
  return dw_wgrad_march_slots(N, gd[0], gd[1], gd[2], C, K, stride, dtype);
}

static bool wg_vec_ok(int C, int K, int dtype) {
  const int epv = dtype == PYTC_BF16 ? 8 : 4;
  return K == 3 && C % epv == 0 && C / epv <= 256 && tuning_get("dw_wgrad_vec", 1) != 0;
}
static void make_wg_vec(DwWg& q, long& rps, int N, const int32_t* gd, const int32_t* xd, int C, int K, int stride, int dtype) {
  q.N = N; q.Dg = gd[0]; q.Hg = gd[1]; q.Wg = gd[2]; q.Dx = xd[0]; q.Hx = xd[1]; q.Wx = xd[2];
  q.C = C; q.K = K; q.stride = stride; q.pad = K / 2; q.lpv = q.vs = q.iters = 0;
  const int PL = 256 / (C / (dtype == PYTC_BF16 ? 8 : 4));
  const long vg = (long)q.Dg * q.Hg * q.Wg;
  // positions per lane: 16 where that still gives the chip >= 1024 workgroups, down to 4 below (a 14^3 x 256 launch had 264 workgroups whose
  // lanes walked 16 positions, one L2 round trip each: 50 us for 11 MB; knob dw_wgrad_vec_ppl forces a value)
  long ppl = tuning_get("dw_wgrad_vec_ppl", 0);
  if (ppl <= 0) {
    ppl = vg * N * 3 / ((long)PL * 1024);
    ppl = ppl < 4 ? 4 : (ppl > 16 ? 16 : ppl);
  }
  long sl = (vg + (long)PL * ppl - 1) / ((long)PL * ppl);
  q.slots = (int)(sl < 1 ? 1 : (sl > 1024 ? 1024 : sl));
  rps = (vg + q.slots - 1) / q.slots;
}

static bool make_wg(DwWg& q, int N, const int32_t* gd, const int32_t* xd, int C, int K, int stride, int dtype, int& vec) {
  int maxv = dtype == PYTC_BF16 ? 4 : 2;
  vec = 0;
  for (int v = maxv; v >= 1; v >>= 1)
    if (C % v == 0 && C / v <= 256) { vec = v; break; }
  if (!vec) return false;
  q.N = N; q.Dg = gd[0]; q.Hg = gd[1]; q.Wg = gd[2]; q.Dx = xd[0]; q.Hx = xd[1]; q.Wx = xd[2];
  q.C = C; q.K = K; q.stride = stride; q.pad = K / 2;
  q.lpv = C / vec; q.vs = 256 / q.lpv;
  long vg = (long)q.Dg * q.Hg * q.Wg;
  long it = vg / ((long)q.vs * 64);
  q.iters = (int)(it < 1 ? 1 : (it > 256 ? 256 : it));
  q.slots = (int)((vg + (long)q.vs * q.iters - 1) / ((long)q.vs * q.iters));
  return true;
}

extern "C" int pytc_dw_wgrad_slots(int N, const int32_t* gdims, const int32_t* xdims, int C, int K, int stride, int dtype) {
  DwWg q; int vec;
  const int ms = march_slots(N, gdims, xdims, C, K, stride, dtype);
  if (ms > 0) return ms;
  if (wg_vec_ok(C, K, dtype)) { long rps; make_wg_vec(q, rps, N, gdims, xdims, C, K, stride, dtype); return q.slots * N; }
  if (!make_wg(q, N, gdims, xdims, C, K, stride, dtype, vec)) return -1;
  return q.slots * N;
}

template <typename T, int VEC>
static int launch_dwwg(const void* g, const void* x, float* dWp, float* dbp, const DwWg& q, hipStream_t s) {
  dim3 grid(q.slots, q.N), block(256);
  size_t lds = (size_t)q.vs * q.C * sizeof(float);
  switch (q.K) {
    case 3: hipLaunchKernelGGL((dw_wgrad_kernel<T, VEC, 3>), grid, block, lds, s, (const T*)g, (const T*)x, dWp, dbp, q); break;
    case 5: hipLaunchKernelGGL((dw_wgrad_kernel<T, VEC, 5>), grid, block, lds, s, (const T*)g, (const T*)x, dWp, dbp, q); break;
    case 7: hipLaunchKernelGGL((dw_wgrad_kernel<T, VEC, 7>), grid, block, lds, s, (const T*)g, (const T*)x, dWp, dbp, q); break;
    default: set_error("dw_wgrad: unsupported kernel size %d", q.K); return PYTC_ERR_UNSUPPORTED;
  }
  return PYTC_OK;
}

static int dw_wgrad_impl(const void* g, const void* x, float* dW, float* db, float* workspace, int N,
                         const int32_t* gdims, const int32_t* xdims, int C, int K, int stride, int dtype,
                         void* stream, bool reduce, int* slots_out) {
  PYTC_REQUIRE(g && x && (dW || !reduce) && workspace && gdims && xdims, "dw_wgrad: null pointer");
  const int ms = march_slots(N, gdims, xdims, C, K, stride, dtype);
  if (ms > 0) {
    const long nWm = 27L * C;
    float* dWm = workspace;
    float* dbm = workspace + (long)ms * nWm;
    hipStream_t sm = (hipStream_t)stream;
    dw_wgrad_march_launch(g, x, dWm, db ? dbm : nullptr, N, gdims[0], gdims[1], gdims[2], C, dtype, sm);
    if (slots_out) *slots_out = ms;
    if (reduce) reduce_slots_pair(dWm, dW, nWm, dbm, db, (long)C, ms, sm);
    PYTC_LAUNCH_CHECK("dw_wgrad");
    return PYTC_OK;
  }
  DwWg q; int vec;
  if (wg_vec_ok(C, K, dtype)) {
    long rps;
    make_wg_vec(q, rps, N, gdims, xdims, C, K, stride, dtype);
    const int total = q.slots * N;
    const long nWv = 27L * C;
    float* dWv = workspace;
    float* dbv = workspace + (long)total * nWv;
    hipStream_t sv = (hipStream_t)stream;
    q.kz_inner = tuning_get("dw_wgrad_vec_kz_inner", 1);
    dim3 grid(q.slots, N, 3), block(256);
    if (q.kz_inner) grid = dim3((unsigned)(q.slots * N * 3), 1, 1);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(dw_wgrad_vec_kernel<bf16_t>, grid, block, 0, sv, (const bf16_t*)g, (const bf16_t*)x, dWv, db ? dbv : nullptr, q, rps),
               hipLaunchKernelGGL(dw_wgrad_vec_kernel<float>, grid, block, 0, sv, (const float*)g, (const float*)x, dWv, db ? dbv : nullptr, q, rps),
               "dw_wgrad")
    if (slots_out) *slots_out = total;
    if (reduce) reduce_slots_pair(dWv, dW, nWv, dbv, db, (long)C, total, sv);
    PYTC_LAUNCH_CHECK("dw_wgrad");
    return PYTC_OK;
  }
  if (!make_wg(q, N, gdims, xdims, C, K, stride, dtype, vec)) { set_error("dw_wgrad: unsupported channel count %d", C); return PYTC_ERR_UNSUPPORTED; }
  PYTC_REQUIRE((size_t)q.vs * C * sizeof(float) <= 64 * 1024, "dw_wgrad: scratch too large");
  const int total_slots = q.slots * N;
  const long nW = (long)K * K * K * C;
  float* dWp = workspace;
  float* dbp = workspace + (long)total_slots * nW;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (dtype == PYTC_BF16) rc = vec == 4 ? launch_dwwg<bf16_t, 4>(g, x, dWp, dbp, q, s) : vec == 2 ? launch_dwwg<bf16_t, 2>(g, x, dWp, dbp, q, s) : launch_dwwg<bf16_t, 1>(g, x, dWp, dbp, q, s);
  else rc = vec == 2 ? launch_dwwg<float, 2>(g, x, dWp, dbp, q, s) : launch_dwwg<float, 1>(g, x, dWp, dbp, q, s);
  if (rc != PYTC_OK) return rc;
  if (slots_out) *slots_out = total_slots;
  if (reduce) reduce_slots_pair(dWp, dW, nW, dbp, db, (long)C, total_slots, s);
  PYTC_LAUNCH_CHECK("dw_wgrad");
  return PYTC_OK;
}

extern "C" int pytc_dw_wgrad(const void* g, const void* x, float* dW, float* db, float* workspace, int N,
                             const int32_t* gdims, const int32_t* xdims, int C, int K, int stride, int dtype,
                             void* stream) {
  return dw_wgrad_impl(g, x, dW, db, workspace, N, gdims, xdims, C, K, stride, dtype, stream, true, nullptr);
}

extern "C" int pytc_dw_wgrad_partial(const void* g, const void* x, float* workspace, int want_db, int N, const int32_t* gdims,
                                     const int32_t* xdims, int C, int K, int stride, int dtype, int* slots_out, void* stream) {
  PYTC_REQUIRE(slots_out, "dw_wgrad_partial: null slots_out");
  return dw_wgrad_impl(g, x, nullptr, want_db ? workspace : nullptr, workspace, N, gdims, xdims, C, K, stride, dtype, stream,
                       false, slots_out);
}

extern "C" int pytc_norm_bwd(const void* dtn, const void* t, const float* mean_rstd, const float* gamma,
                             float* stats_ws, float* s_out, void* dt, int N, int64_t rows, float count, int C,
                             int dtype, void* stream) {
  PYTC_REQUIRE(dtn && t && mean_rstd && stats_ws && s_out && dt, "norm_bwd: null pointer");
  const int slots = colstats_slots(rows);
  const long rps = (rows + slots - 1) / slots;
  const int Cw = C < 256 ? C : 256;
  size_t lds = (size_t)(256 / Cw) * 2 * Cw * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(slots, N), block(256);
  const bool vec = dtype == PYTC_BF16 ? (C % 8 == 0) : (C % 4 == 0);
  if (vec) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((colstats_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)dtn, (const bf16_t*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               hipLaunchKernelGGL((colstats_kernel<float, 1>), grid, block, 0, s, (const float*)dtn, (const float*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               "norm_bwd")
  } else {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(norm_bwd_stats_kernel<bf16_t>, grid, block, lds, s, (const bf16_t*)dtn, (const bf16_t*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               hipLaunchKernelGGL(norm_bwd_stats_kernel<float>, grid, block, lds, s, (const float*)dtn, (const float*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               "norm_bwd")
  }
  // reduce slots per sample: stats_ws [N][slots][2][C] -> s_out [N][2][C]
  hipLaunchKernelGGL(reduce_slots_batched_kernel, dim3(ceil_div(2L * C, 16), N), dim3(256), 0, s, stats_ws, s_out, 2L * C, slots);
  const long total = (long)N * rows * C;
  if (vec) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(norm_bwd_apply_vec_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)dtn, (const bf16_t*)t, mean_rstd, gamma, s_out, 1.0f / count, (bf16_t*)dt, (long)rows, C, rps),
               hipLaunchKernelGGL(norm_bwd_apply_vec_kernel<float>, grid, block, 0, s, (const float*)dtn, (const float*)t, mean_rstd, gamma, s_out, 1.0f / count, (float*)dt, (long)rows, C, rps),
               "norm_bwd")
  } else {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(norm_bwd_apply_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dtn, (const bf16_t*)t, mean_rstd, gamma, s_out, 1.0f / count, (bf16_t*)dt, (long)rows, C, total),
               hipLaunchKernelGGL(norm_bwd_apply_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dtn, (const float*)t, mean_rstd, gamma, s_out, 1.0f / count, (float*)dt, (long)rows, C, total),
               "norm_bwd")
  }
  PYTC_LAUNCH_CHECK("norm_bwd");
  return PYTC_OK;
}

/* statistics pass only: s_out [N][2][C] = (sum d, sum d * xhat) per (sample, channel) */
extern "C" int pytc_norm_bwd_stats(const void* dtn, const void* t, const float* mean_rstd, float* stats_ws, float* s_out,
                                   int N, int64_t rows, int C, int dtype, void* stream) {
  PYTC_REQUIRE(dtn && t && mean_rstd && stats_ws && s_out, "norm_bwd_stats: null pointer");
  const int slots = colstats_slots(rows);
  const long rps = (rows + slots - 1) / slots;
  const int Cw = C < 256 ? C : 256;
  size_t lds = (size_t)(256 / Cw) * 2 * Cw * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(slots, N), block(256);
  const bool vec = dtype == PYTC_BF16 ? (C % 8 == 0) : (C % 4 == 0);
  if (vec) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((colstats_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)dtn, (const bf16_t*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               hipLaunchKernelGGL((colstats_kernel<float, 1>), grid, block, 0, s, (const float*)dtn, (const float*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               "norm_bwd_stats")
  } else {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(norm_bwd_stats_kernel<bf16_t>, grid, block, lds, s, (const bf16_t*)dtn, (const bf16_t*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               hipLaunchKernelGGL(norm_bwd_stats_kernel<float>, grid, block, lds, s, (const float*)dtn, (const float*)t, mean_rstd, stats_ws, (long)rows, C, slots, rps),
               "norm_bwd_stats")
  }
  hipLaunchKernelGGL(reduce_slots_batched_kernel, dim3(ceil_div(2L * C, 16), N), dim3(256), 0, s, stats_ws, s_out, 2L * C, slots);
  PYTC_LAUNCH_CHECK("norm_bwd_stats");
  return PYTC_OK;
}

extern "C" int pytc_norm_bwd_ws_elems(int N, int64_t rows, int C) {
  const int slots = colstats_slots(rows);
  return (int)((long)N * slots * 2 * C);
}

static int dwconv3d_bwd_data_impl(const void* dy, const float* w, const void* addend, void* dx, int N, const int32_t* xdims,
                                  const int32_t* ydims, int C, int K, int stride, int dtype, void* stream);

extern "C" int pytc_dwconv3d_bwd_data(const void* dy, const float* w, void* dx, int N, const int32_t* xdims,
                                      const int32_t* ydims, int C, int K, int stride, int dtype, void* stream) {
  return dwconv3d_bwd_data_impl(dy, w, nullptr, dx, N, xdims, ydims, C, K, stride, dtype, stream);
}

/* dx = conv^T(dy) + addend (addend shaped like dx; 16-byte channel groups only) */
extern "C" int pytc_dwconv3d_bwd_data_add(const void* dy, const float* w, const void* addend, void* dx, int N, const int32_t* xdims,
                                          const int32_t* ydims, int C, int K, int stride, int dtype, void* stream) {
  PYTC_REQUIRE(addend && C % (dtype == PYTC_BF16 ? 8 : 4) == 0, "dwconv3d_bwd_data_add: addend and 16-byte channel groups required (C = %d)", C);
  return dwconv3d_bwd_data_impl(dy, w, addend, dx, N, xdims, ydims, C, K, stride, dtype, stream);
}

static int dwconv3d_bwd_data_impl(const void* dy, const float* w, const void* addend, void* dx, int N, const int32_t* xdims,
                                  const int32_t* ydims, int C, int K, int stride, int dtype, void* stream) {
  PYTC_REQUIRE(dy && w && dx && xdims && ydims, "dwconv3d_bwd_data: null pointer");
  DwBd g;
  g.D = xdims[0]; g.H = xdims[1]; g.W = xdims[2]; g.Do = ydims[0]; g.Ho = ydims[1]; g.Wo = ydims[2];
  g.C = C; g.K = K; g.stride = stride; g.pad = K / 2;
  const long total = (long)N * g.D * g.H * g.W * C;
  hipStream_t s = (hipStream_t)stream;
  const int epv = dtype == PYTC_BF16 ? 8 : 4;
  if (C % epv == 0) {
    const long tv = total / epv;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(dwconv_bwd_data_vec_kernel<bf16_t>, dim3(ceil_div(tv, 256)), dim3(256), 0, s, (const bf16_t*)dy, w, (bf16_t*)dx, g, tv, (const bf16_t*)addend),
               hipLaunchKernelGGL(dwconv_bwd_data_vec_kernel<float>, dim3(ceil_div(tv, 256)), dim3(256), 0, s, (const float*)dy, w, (float*)dx, g, tv, (const float*)addend),
               "dwconv3d_bwd_data")
    PYTC_LAUNCH_CHECK("dwconv3d_bwd_data");
    return PYTC_OK;
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(dwconv_bwd_data_kernel<bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const bf16_t*)dy, w, (bf16_t*)dx, g, total),
             hipLaunchKernelGGL(dwconv_bwd_data_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)dy, w, (float*)dx, g, total),
             "dwconv3d_bwd_data")
  PYTC_LAUNCH_CHECK("dwconv3d_bwd_data");
  return PYTC_OK;
}
