// Backward kernels of the dense-convolution (RSUNet) training step.  Correctness-first: VALU arithmetic, two-stage
// deterministic reductions (per-workgroup partial slots -> reduce in slot order), no float atomics.
//   conv3d weight gradient (one TN "GEMM" per tap over the voxel rows, shifted operand, zero padding),
//   activation derivative through the norm affine, general norm backward apply (any statistics group),
//   max-pool backward (first maximum in scan order, like torch), anisotropic depthwise conv (backward of the fixed
//   bilinear transposed-conv upsampling).
// Data gradients of the dense convs reuse the forward implicit-GEMM kernel with flipped / transposed weights.
#include "pytc_common.h"
#include "colstats.h"

namespace pytc {

__device__ __forceinline__ float act_fwd(float t, int act, float prm) {
  if (act == PYTC_ACT_RELU) return fmaxf(t, 0.f);
  if (act == PYTC_ACT_LEAKY) return t > 0.f ? t : prm * t;
  if (act == PYTC_ACT_ELU) return t > 0.f ? t : prm * (__expf(t) - 1.f);
  return t;
}
__device__ __forceinline__ float act_der(float t, int act, float prm) {
  if (act == PYTC_ACT_RELU) return t > 0.f ? 1.f : 0.f;
  if (act == PYTC_ACT_LEAKY) return t > 0.f ? 1.f : prm;
  if (act == PYTC_ACT_ELU) return t > 0.f ? 1.f : prm * __expf(t);
  return 1.f;
}

// ---- conv3d weight gradient ------------------------------------------------------------------------------------------
// dWp[slot][tap][o][k] = sum_{rows of slot} dY[r][o] * A[r + shift(tap)][k], A = conv input (already activated), zero
// outside the volume.  workgroup = (row slot, 64x64 (o,k) tile, tap); rows staged 32 at a time; thread = 4x4 block.
struct CwGeom { int D, H, W, kd, kh, kw; };
constexpr int CW_TO = 64, CW_TK = 64, CW_TR = 32;
template <typename T>
__global__ void __launch_bounds__(256)
conv3d_wgrad_kernel(const T* __restrict__ a, const T* __restrict__ dy, float* __restrict__ dWp, long rows_total,
                    CwGeom g, int C_in, int C_out, long rows_per_slot, int slots) {
  __shared__ float sa[CW_TR][CW_TK + 1];
  __shared__ float sd[CW_TR][CW_TO + 1];
  const int slot = blockIdx.x, tap = blockIdx.z;
  const int tiles_k = (C_in + CW_TK - 1) / CW_TK;
  const int o_base = (blockIdx.y / tiles_k) * CW_TO, k_base = (blockIdx.y % tiles_k) * CW_TK;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int tz_ = tap / (g.kh * g.kw), ty_ = (tap / g.kw) % g.kh, tx_ = tap % g.kw;
  const int dz = tz_ - g.kd / 2, dyy = ty_ - g.kh / 2, dx = tx_ - g.kw / 2;
  const long vol = (long)g.D * g.H * g.W;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const long r_begin = (long)slot * rows_per_slot;
  const long r_end = r_begin + rows_per_slot < rows_total ? r_begin + rows_per_slot : rows_total;
  for (long r0 = r_begin; r0 < r_end; r0 += CW_TR) {
    for (int i = threadIdx.x; i < CW_TR * CW_TK; i += 256) {
      const int rr = i / CW_TK, kk = i % CW_TK;
      const long r = r0 + rr;
      float v = 0.f;
      if (r < r_end && k_base + kk < C_in) {
        const long rem = r % vol;
        const int x = (int)(rem % g.W), y = (int)((rem / g.W) % g.H), z = (int)(rem / ((long)g.W * g.H));
        const int sz = z + dz, sy = y + dyy, sx = x + dx;
        if (sz >= 0 && sz < g.D && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W)
          v = to_f32<T>(a[(r + ((long)dz * g.H + dyy) * g.W + dx) * C_in + k_base + kk]);
      }
      sa[rr][kk] = v;
    }
    for (int i = threadIdx.x; i < CW_TR * CW_TO; i += 256) {
      const int rr = i / CW_TO, oo = i % CW_TO;
      const long r = r0 + rr;
      sd[rr][oo] = (r < r_end && o_base + oo < C_out) ? to_f32<T>(dy[r * C_out + o_base + oo]) : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < CW_TR; ++rr) {
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
    if (o >= C_out) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k_base + tx * 4 + j;
      if (k < C_in) dWp[(((long)slot * taps + tap) * C_out + o) * C_in + k] = acc[i][j];
    }
  }
}

// ---- conv3d weight gradient on MFMA (bf16, kh = kw = KP in {3, 1}) -----------------------------------------------------
// The reduction runs over voxels; a wave's unit of work is one 32-voxel segment of an x line.  It stages the dY rows
// of the segment and, for the kernel plane dz of its workgroup, the KP neighbouring A lines (y-1, y, y+1) with the x
// halo (zero outside the volume) into a wave-private LDS image, then forms, per in-plane tap (dy, dx), the 16x16x32
// products  dW[tap] += dY^T * A(shifted)  - the x shift of a tap is a row offset into the staged line, so one staged
// image serves all KP*KP taps.  Fragments come from gfx950's LDS transpose read as in pw_wgrad_mfma_kernel (same
// row <-> k-slot map for both operands).  No workgroup barrier in the loop; the next unit's global loads are in flight
// during the MFMAs.  Workgroup = (row slot, (o, k) tile, dz); the blocks of one slot are dispatched back to back onto
// ONE XCD (linear id -> XCD = id % 8) so the re-reads of the slot's rows by its (tile, dz) blocks hit that XCD's L2.
// RAG_O / RAG_K: a channel count below 16 on that side (1-channel input conv, 3-channel heads): the rows are not
// 16-byte aligned, so the operand is gathered element-wise into a zero-padded 16-channel LDS tile (the op is
// HBM-bound there; the padded MFMA columns are free).  Cross-wave sum in fixed order, per-slot partials, slot-ordered
// reduction: deterministic.
template <int MT, int NT, int KP, bool RAG_O, bool RAG_K>
__global__ void __launch_bounds__(256, (MT * NT * KP * KP >= 36 ? 2 : 3))
conv3d_wgrad_mfma_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ dy, float* __restrict__ dWp, int N,
                         CwGeom g, int C_in, int C_out, long units_per_slot, int slots, int tiles) {
  static_assert(!RAG_O || MT == 1, "ragged output side uses one 16-channel tile");
  static_assert(!RAG_K || NT == 1, "ragged input side uses one 16-channel tile");
  constexpr int BM = MT * 16, BN = NT * 16;
  constexpr int SG = BM * 2 + 32, SX = BN * 2 + 32;        // LDS row pitch in bytes
  constexpr int PAD = KP / 2, AR = 32 + 2 * PAD;            // staged rows of an A line: 32 + the x halo
  constexpr int TP = KP * KP;
  constexpr int WAVE_BYTES = 32 * SG + KP * AR * SX;
  constexpr int RED_BYTES = TP * BM * BN * 4;
  constexpr int LDS_BYTES = 4 * WAVE_BYTES > RED_BYTES ? 4 * WAVE_BYTES : RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  typedef unsigned int q4_t __attribute__((ext_vector_type(4)));      // (arrays of the uint4 STRUCT end up in scratch memory)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* lg = lds + wave * WAVE_BYTES;
  unsigned char* la = lg + 32 * SG;
  // linear block id -> (slot, tile, dz): slot % 8 = XCD, the slot's blocks consecutive on that XCD
  const int T = tiles * g.kd;
  const long b = blockIdx.x;
  const int slot = (int)((b / 8 / T) * 8 + (b % 8));
  if (slot >= slots) return;
  const int t = (int)((b / 8) % T);
  const int dzi = t % g.kd, tile = t / g.kd;
  const int tiles_k = RAG_K ? 1 : (C_in + BN - 1) / BN;      // the last tile may be partly filled (C % 8 == 0 is all that is asked)
  const int o_base = (tile / tiles_k) * BM, k_base = (tile % tiles_k) * BN;
  const int dz = dzi - g.kd / 2;
  const int nseg = (g.W + 31) / 32;
  const long units = (long)N * g.D * g.H * nseg;
  const long u_begin = (long)slot * units_per_slot;
  const long u_end = u_begin + units_per_slot < units ? u_begin + units_per_slot : units;

  constexpr int CHG = BM / 8, RG = 64 / CHG, ITG = 32 / RG;   // dY: 16-B chunks per row, rows per load, loads
  constexpr int CHX = BN / 8, ITA = (AR * CHX + 63) / 64;     // A line: chunk loads per lane
  const int g_row = lane / CHG, g_chunk = lane % CHG;
  q4_t rg[ITG], ra[KP][ITA];
  const q4_t zero4 = {0u, 0u, 0u, 0u};
  if (RAG_O || RAG_K) {                 // zero the padded channels once; the gather only rewrites the real ones
    for (int i = lane; i < WAVE_BYTES / 16; i += 64) reinterpret_cast<q4_t*>(lg)[i] = zero4;
  }
  // unit u -> (column, y) with y fastest: a wave walks DOWN a column of x segments, so of the KP A lines of a unit only the
  // newest is loaded (the others already sit in the LDS ring from the previous unit); full = first unit of a run / column
  auto fetch = [&](long u, bool full) {
    const int y = (int)(u % g.H);
    const long col = u / g.H;
    const int xs = (int)(col % nseg);
    const long nz = col / nseg;                     // n * D + z
    const int z = (int)(nz % g.D);
    const long line = nz * g.H + y;
    const int x0 = xs * 32;
    if (RAG_O) {                        // element e = j*64 + lane of the segment's 32 x C_out values
      const unsigned short* dyl = reinterpret_cast<const unsigned short*>(dy) + line * g.W * (long)C_out;
      u16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = j * 64 + lane;
        const int row = e / C_out, x = x0 + row;
        v[j] = (row < 32 && x < g.W) ? dyl[(long)x0 * C_out + e] : (unsigned short)0;
      }
      rg[0] = __builtin_bit_cast(q4_t, v);
    } else {
      const bf16_t* dyl = dy + line * g.W * (long)C_out + o_base + g_chunk * 8;
      const bool och = o_base + g_chunk * 8 < C_out;        // 8-channel pieces beyond C_out (partly filled last tile): zeros
#pragma unroll
      for (int it = 0; it < ITG; ++it) {
        const int x = x0 + it * RG + g_row;
        rg[it] = (x < g.W && och) ? *reinterpret_cast<const q4_t*>(dyl + (long)x * C_out) : zero4;
      }
    }
    const int sz = z + dz;
#pragma unroll
    for (int l = 0; l < KP; ++l) {
      if (!full && l != KP - 1) continue;
      const int sy = y + l - PAD;
      const bool ok = sz >= 0 && sz < g.D && sy >= 0 && sy < g.H;         // wave-uniform
      if (RAG_K) {
        const unsigned short* al = reinterpret_cast<const unsigned short*>(a) + (line + (long)dz * g.H + (l - PAD)) * g.W * (long)C_in;
        u16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = j * 64 + lane;
          const int row = e / C_in, x = x0 - PAD + row;
          v[j] = (ok && row < AR && x >= 0 && x < g.W) ? al[((long)x0 - PAD) * C_in + e] : (unsigned short)0;
        }
        ra[l][0] = __builtin_bit_cast(q4_t, v);
      } else {
        const bf16_t* al = a + (line + (long)dz * g.H + (l - PAD)) * g.W * (long)C_in + k_base;
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
          const int c = it * 64 + lane;
          const int row = c / CHX, chunk = c % CHX;
          const int x = x0 - PAD + row;
          ra[l][it] = (ok && row < AR && x >= 0 && x < g.W && k_base + chunk * 8 < C_in)
                          ? *reinterpret_cast<const q4_t*>(al + (long)x * C_in + chunk * 8) : zero4;
        }
      }
    }
  };
  auto ring = [&](int y, int l) { return (y + l - PAD + KP) % KP; };       // LDS slot of line y + l - PAD
  auto stage = [&](int y, bool full) {
    if (RAG_O) {
      const u16x8 v = __builtin_bit_cast(u16x8, rg[0]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = j * 64 + lane;
        const int row = e / C_out, col = e % C_out;
        if (row < 32) *reinterpret_cast<unsigned short*>(lg + row * SG + col * 2) = v[j];
      }
    } else {
#pragma unroll
      for (int it = 0; it < ITG; ++it) *reinterpret_cast<q4_t*>(lg + (it * RG + g_row) * SG + g_chunk * 16) = rg[it];
    }
#pragma unroll
    for (int l = 0; l < KP; ++l) {
      if (!full && l != KP - 1) continue;
      const int sl = ring(y, l);
      if (RAG_K) {
        const u16x8 v = __builtin_bit_cast(u16x8, ra[l][0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = j * 64 + lane;
          const int row = e / C_in, col = e % C_in;
          if (row < AR) *reinterpret_cast<unsigned short*>(la + (sl * AR + row) * SX + col * 2) = v[j];
        }
      } else {
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
          const int c = it * 64 + lane;
          const int row = c / CHX, chunk = c % CHX;
          if (row < AR) *reinterpret_cast<q4_t*>(la + (sl * AR + row) * SX + chunk * 16) = ra[l][it];
        }
      }
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

  f32x4_t acc[TP][MT][NT];
#pragma unroll
  for (int tp = 0; tp < TP; ++tp)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[tp][m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // the slot's units in four contiguous runs, one per wave
  const long quarter = (u_end - u_begin + 3) / 4;
  const long w_begin = u_begin + wave * quarter;
  const long w_end = w_begin + quarter < u_end ? w_begin + quarter : u_end;
  long u = w_begin;
  if (u < w_end) fetch(u, true);
  for (; u < w_end; ++u) {
    const int y = (int)(u % g.H);
    stage(y, u == w_begin || y == 0);
    if (u + 1 < w_end) fetch(u + 1, ((u + 1) % g.H) == 0);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    bf16x8_t fa[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) fa[m] = frag(lg, SG, 0, m);
#pragma unroll
    for (int l = 0; l < KP; ++l)
#pragma unroll
      for (int dx = 0; dx < KP; ++dx) {
        bf16x8_t fb[NT];
        const int sl = ring(y, l);
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = frag(la, SX, sl * AR + dx, n);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[l * KP + dx][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[n], acc[l * KP + dx][m][n], 0, 0, 0);
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
      for (int tp = 0; tp < TP; ++tp)
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
      for (int tp = 0; tp < TP; ++tp)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[tp][m][n][i] += red[(tp * BM + m * 16 + mg + i) * BN + n * 16 + nn];
    }
  }
  if (wave == 0) {
    const int taps = g.kd * TP;
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int o = o_base + m * 16 + mg + i;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int k = k_base + n * 16 + nn;
            if (o < C_out && k < C_in)
              dWp[(((long)slot * taps + dzi * TP + tp) * C_out + o) * C_in + k] = acc[tp][m][n][i];
          }
        }
  }
}

// batched form of reduce_slots2_kernel: blockIdx.y = sample; part [N][slots][n] -> out [N][n], same summation tree
__global__ void __launch_bounds__(256)
reduce_slots_batched_rs_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int slots) {
  __shared__ float sm[16][17];
  part += (long)blockIdx.y * slots * n;
  out += (long)blockIdx.y * n;
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

// out[i] = sum_s part[s][i] in a fixed tree: 16 elements x 16 slot lanes per workgroup (lane j adds slots j, j+16, ...
// in order, then the 16 lane sums in lane order)
__global__ void __launch_bounds__(256)
reduce_slots2_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int slots) {
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

// ---- activation derivative through the norm affine: dt = da * act'(a*x + b);  dp = da * min(t, 0) (PReLU weight) ----
template <typename T>
__global__ void __launch_bounds__(256)
act_bwd_kernel(const T* __restrict__ da, const T* __restrict__ x, const float* __restrict__ ab, T* __restrict__ dt,
               T* __restrict__ dp, long rows, int C, long total, int act, float prm) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long n = i / (rows * C);
    float t = to_f32<T>(x[i]);
    if (ab) t = to_f32<T>(from_f32<T>(fmaf(t, ab[(n * 2 + 0) * C + c], ab[(n * 2 + 1) * C + c]))) ;
    const float d = to_f32<T>(da[i]);
    dt[i] = from_f32<T>(d * act_der(t, act, prm));
    if (dp) dp[i] = from_f32<T>(t < 0.f ? d * t : 0.f);
  }
}

// 16-byte forms of act_bwd / norm_bwd_apply_general (C % VEC == 0): one thread = VEC consecutive channels of one voxel; the
// element-wise arithmetic is the scalar kernels', so results are bit-identical (A/B: pytc_set_tuning("elementwise_vec", 0)).
template <typename T>
__global__ void __launch_bounds__(256)
act_bwd_vec_kernel(const T* __restrict__ da, const T* __restrict__ x, const float* __restrict__ ab, T* __restrict__ dt,
                   T* __restrict__ dp, long rows, int C, long chunks, int act, float prm) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cq = C / VEC;
  const long per_sample = rows * cq;
  long q = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; q < chunks; q += stride) {
    const int c0 = (int)(q % cq) * VEC;
    const long n = q / per_sample;
    float xv[VEC], dv[VEC], o[VEC], pv[VEC];
    VecIO<T, VEC>::load(x + q * VEC, xv);
    VecIO<T, VEC>::load(da + q * VEC, dv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = xv[j];
      if (ab) t = to_f32<T>(from_f32<T>(fmaf(t, ab[(n * 2 + 0) * C + c0 + j], ab[(n * 2 + 1) * C + c0 + j])));
      o[j] = dv[j] * act_der(t, act, prm);
      pv[j] = t < 0.f ? dv[j] * t : 0.f;
    }
    VecIO<T, VEC>::store(dt + q * VEC, o);
    if (dp) VecIO<T, VEC>::store(dp + q * VEC, pv);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
norm_bwd_apply_general_vec_kernel(const T* __restrict__ d, const T* __restrict__ x, const float* __restrict__ mr,
                                  const float* __restrict__ gamma, const float* __restrict__ M, T* __restrict__ dx,
                                  long rows, int C, long chunks) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cq = C / VEC;
  const long per_sample = rows * cq;
  long q = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; q < chunks; q += stride) {
    const int c0 = (int)(q % cq) * VEC;
    const long n = q / per_sample;
    float xv[VEC], dv[VEC], o[VEC];
    VecIO<T, VEC>::load(x + q * VEC, xv);
    VecIO<T, VEC>::load(d + q * VEC, dv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      const float mean = mr[(n * 2 + 0) * C + c], rstd = mr[(n * 2 + 1) * C + c];
      const float xh = (xv[j] - mean) * rstd;
      const float gd = (gamma ? gamma[c] : 1.f) * dv[j];
      o[j] = rstd * (gd - M[(n * 2 + 0) * C + c] - xh * M[(n * 2 + 1) * C + c]);
    }
    VecIO<T, VEC>::store(dx + q * VEC, o);
  }
}

// ---- activation derivative + norm backward WITHOUT the intermediate dt tensor -----------------------------------------------
// The three passes  act_bwd (dt = da * act'(t)) -> norm_bwd_stats(dt, x) -> norm_bwd_apply_general(dt, x)  move 8 tensor-sized
// units (2 reads + 1 write, 2 reads, 2 reads + 1 write); dt is a function of (da, x) alone, so the statistics pass and the apply
// pass each recompute it in registers from the same two tensors: 5 units, one launch less.  dt is rounded to the storage type
// exactly where the three-pass form stored it: the same dt values enter the sums and dx (only the fp32 summation order of the
// slot reduction differs).  PRELU: third column sum da*min(t,0).
template <typename T>
__device__ __forceinline__ void act_dt_vec(const float (&xv)[16 / sizeof(T)], const float (&dv)[16 / sizeof(T)], const float* a,
                                           const float* b, int act, float prm, float (&dt)[16 / sizeof(T)],
                                           float (&dp)[16 / sizeof(T)]) {
  constexpr int VEC = 16 / (int)sizeof(T);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float t = xv[j];
    if (a) t = to_f32<T>(from_f32<T>(fmaf(t, a[j], b[j])));
    dt[j] = to_f32<T>(from_f32<T>(dv[j] * act_der(t, act, prm)));
    dp[j] = to_f32<T>(from_f32<T>(t < 0.f ? dv[j] * t : 0.f));
  }
}

template <typename T, bool PRELU>
__global__ void __launch_bounds__(256)
act_norm_bwd_stats_kernel(const T* __restrict__ da, const T* __restrict__ x, const float* __restrict__ ab,
                          const float* __restrict__ mr, float* __restrict__ stats, float* __restrict__ pstats, long rows, int C,
                          int slots, long rows_per_slot, int act, float prm) {
  constexpr int EPV = 16 / (int)sizeof(T), W = PRELU ? 3 : 2;
  __shared__ float lds[W * 256 * EPV];                 // [row lane][W][Cw * EPV]
  const int n = blockIdx.y, slot = blockIdx.x;
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
  const T* dn = da + (long)n * rows * C;
  const T* xn = x + (long)n * rows * C;
  const int chunks = C / EPV;
  for (int k0 = 0; k0 < chunks; k0 += 256) {
    const int Cw = (chunks - k0) < 256 ? (chunks - k0) : 256;
    const int RL = 256 / Cw;
    const int ck = threadIdx.x % Cw, rl = threadIdx.x / Cw;
    const int c = (k0 + ck) * EPV;
    if (rl < RL) {
      float s1[EPV], s2[EPV], s3[EPV], mean[EPV], rstd[EPV], av[EPV], bv[EPV];
#pragma unroll
      for (int i = 0; i < EPV; ++i) {
        s1[i] = 0.f; s2[i] = 0.f; s3[i] = 0.f;
        mean[i] = mr[((long)n * 2 + 0) * C + c + i];
        rstd[i] = mr[((long)n * 2 + 1) * C + c + i];
        av[i] = ab ? ab[((long)n * 2 + 0) * C + c + i] : 1.f;
        bv[i] = ab ? ab[((long)n * 2 + 1) * C + c + i] : 0.f;
      }
#pragma unroll 2
      for (long r = r0 + rl; r < r1; r += RL) {
        float dv[EPV], xv[EPV], dt[EPV], dp[EPV];
        VecIO<T, EPV>::load(dn + r * C + c, dv);
        VecIO<T, EPV>::load(xn + r * C + c, xv);
        act_dt_vec<T>(xv, dv, ab ? av : nullptr, bv, act, prm, dt, dp);
#pragma unroll
        for (int i = 0; i < EPV; ++i) {
          s1[i] += dt[i];
          s2[i] = fmaf(dt[i], (xv[i] - mean[i]) * rstd[i], s2[i]);
          if (PRELU) s3[i] += dp[i];
        }
      }
#pragma unroll
      for (int i = 0; i < EPV; ++i) {
        lds[((rl * W + 0) * Cw + ck) * EPV + i] = s1[i];
        lds[((rl * W + 1) * Cw + ck) * EPV + i] = s2[i];
        if (PRELU) lds[((rl * W + 2) * Cw + ck) * EPV + i] = s3[i];
      }
    }
    __syncthreads();
    const int width = Cw * EPV;
    for (int i = threadIdx.x; i < W * width; i += 256) {
      const int which = i / width, e = i % width;
      float acc = 0.f;
      for (int q = 0; q < RL; ++q) acc += lds[(q * W + which) * width + e];
      if (which < 2) stats[(((long)n * slots + slot) * 2 + which) * C + k0 * EPV + e] = acc;
      else pstats[((long)n * slots + slot) * C + k0 * EPV + e] = acc;
    }
    __syncthreads();
  }
}

// apply pass: workgroup = (row slot, sample), lane = (16-byte channel chunk, row lane); the seven per-channel coefficients live
// in registers for the whole slot (a grid-stride form re-loaded them per element: 56 scalar loads around two 16-byte ones)
template <typename T>
__global__ void __launch_bounds__(256)
act_norm_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ x, const float* __restrict__ ab,
                          const float* __restrict__ mr, const float* __restrict__ gamma, const float* __restrict__ M,
                          T* __restrict__ dx, long rows, int C, long rows_per_slot, int act, float prm, int C_gamma) {
  // C_gamma: entries of gamma (the norm's own channels); the channels from there on are alignment padding and take gamma = 0
  constexpr int EPV = 16 / (int)sizeof(T);
  const int n = blockIdx.y, slot = blockIdx.x;
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < rows ? r0 + rows_per_slot : rows;
  const long base = (long)n * rows * C;
  const int chunks = C / EPV;
  for (int k0 = 0; k0 < chunks; k0 += 256) {
    const int Cw = (chunks - k0) < 256 ? (chunks - k0) : 256;
    const int RL = 256 / Cw;
    const int ck = threadIdx.x % Cw, rl = threadIdx.x / Cw;
    if (rl >= RL) continue;
    const int c = (k0 + ck) * EPV;
    float mean[EPV], rstd[EPV], g[EPV], m1[EPV], m2[EPV], av[EPV], bv[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) {
      mean[i] = mr[((long)n * 2 + 0) * C + c + i];
      rstd[i] = mr[((long)n * 2 + 1) * C + c + i];
      g[i] = gamma ? (c + i < C_gamma ? gamma[c + i] : 0.f) : 1.f;
      m1[i] = M[((long)n * 2 + 0) * C + c + i];
      m2[i] = M[((long)n * 2 + 1) * C + c + i];
      av[i] = ab ? ab[((long)n * 2 + 0) * C + c + i] : 1.f;
      bv[i] = ab ? ab[((long)n * 2 + 1) * C + c + i] : 0.f;
    }
#pragma unroll 2
    for (long r = r0 + rl; r < r1; r += RL) {
      float dv[EPV], xv[EPV], dt[EPV], dp[EPV], o[EPV];
      VecIO<T, EPV>::load(da + base + r * C + c, dv);
      VecIO<T, EPV>::load(x + base + r * C + c, xv);
      act_dt_vec<T>(xv, dv, ab ? av : nullptr, bv, act, prm, dt, dp);
#pragma unroll
      for (int i = 0; i < EPV; ++i) {
        const float xh = (xv[i] - mean[i]) * rstd[i];
        o[i] = rstd[i] * (g[i] * dt[i] - m1[i] - xh * m2[i]);
      }
      VecIO<T, EPV>::store(dx + base + r * C + c, o);
    }
  }
}

// row slots of the two kernels above: >= 16 rows per lane (a 16-channel tensor has 128 row lanes per workgroup; with the 64-row
// slots of colstats_slots a lane met 3 rows and the launch was all prologue and LDS reduction)
static inline int act_norm_slots(long rows, int C, int epv) {
  const int chunks = C / epv, Cw = chunks < 256 ? chunks : 256, RL = 256 / Cw;
  const long s = rows / ((long)RL * 16), cap = colstats_slots(rows);      // the workspace is sized for colstats_slots
  return (int)(s < 1 ? 1 : (s > cap ? cap : s));
}

// ---- general norm backward apply: dx = rstd * (gamma * d - M1 - xhat * M2), M per (n, c) (group-expanded means) -----
template <typename T>
__global__ void __launch_bounds__(256)
norm_bwd_apply_general_kernel(const T* __restrict__ d, const T* __restrict__ x, const float* __restrict__ mr,
                              const float* __restrict__ gamma, const float* __restrict__ M, T* __restrict__ dx,
                              long rows, int C, long total) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long n = i / (rows * C);
    const float mean = mr[(n * 2 + 0) * C + c], rstd = mr[(n * 2 + 1) * C + c];
    const float xh = (to_f32<T>(x[i]) - mean) * rstd;
    const float gd = (gamma ? gamma[c] : 1.f) * to_f32<T>(d[i]);
    dx[i] = from_f32<T>(rstd * (gd - M[(n * 2 + 0) * C + c] - xh * M[(n * 2 + 1) * C + c]));
  }
}

// ---- max-pool backward (kernel == stride == f, floor): dy goes to the first maximum of its window ------------------
template <typename T>
__global__ void __launch_bounds__(256)
maxpool3d_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int D, int H, int W, int C,
                     int fz, int fy, int fx, int Do, int Ho, int Wo, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long t = i / C;
  const int ox = (int)(t % Wo); t /= Wo;
  const int oy = (int)(t % Ho); t /= Ho;
  const int oz = (int)(t % Do);
  const long n = t / Do;
  const T* xn = x + n * (long)D * H * W * C;
  T* dn = dx + n * (long)D * H * W * C;
  float best = -INFINITY;
  long arg = -1;
  for (int a = 0; a < fz; ++a)
    for (int b = 0; b < fy; ++b)
      for (int e = 0; e < fx; ++e) {
        const long p = (((long)(oz * fz + a) * H + (oy * fy + b)) * W + (ox * fx + e)) * C + c;
        const float v = to_f32<T>(xn[p]);
        if (v > best || arg < 0) { best = v; arg = p; }       // first maximum wins (NaN-free inputs)
      }
  dn[arg] = dy[i];
}

// ---- anisotropic depthwise conv (gather): y[o][c] = sum_k x[o*s - p + k][c] * w[k][c] ------------------------------
struct DwGen { int D, H, W, C, kd, kh, kw, sz, sy, sx, pz, py, px, Do, Ho, Wo; };
template <typename T>
__global__ void __launch_bounds__(256)
dwconv3d_generic_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ w, DwGen g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % g.C);
  long t = i / g.C;
  const int ox = (int)(t % g.Wo); t /= g.Wo;
  const int oy = (int)(t % g.Ho); t /= g.Ho;
  const int oz = (int)(t % g.Do);
  const long n = t / g.Do;
  const T* xn = x + n * (long)g.D * g.H * g.W * g.C;
  float acc = 0.f;
  for (int a = 0; a < g.kd; ++a) {
    const int iz = oz * g.sz - g.pz + a;
    if (iz < 0 || iz >= g.D) continue;
    for (int b = 0; b < g.kh; ++b) {
      const int iy = oy * g.sy - g.py + b;
      if (iy < 0 || iy >= g.H) continue;
      for (int e = 0; e < g.kw; ++e) {
        const int ix = ox * g.sx - g.px + e;
        if (ix < 0 || ix >= g.W) continue;
        acc = fmaf(to_f32<T>(xn[(((long)iz * g.H + iy) * g.W + ix) * g.C + c]), w[((long)(a * g.kh + b) * g.kw + e) * g.C + c], acc);
      }
    }
  }
  y[i] = from_f32<T>(acc);
}

static int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 16384 ? b : 16384); }

// ---- tiny per-layer combiners (one workgroup): replace a dozen small tensor ops per norm layer ------------------------
// s [N][2][C] = (sum d, sum d*xhat) per (sample, channel) -> dbeta[c] = sum_n s[n][0][c], dgamma[c] = sum_n s[n][1][c],
// M [N][2][C] = mean of gamma*s over the statistics group of (n, c): groups > 0: the channel's group within the sample
// (GroupNorm; groups == C: InstanceNorm); groups == 0: all samples of the channel (BatchNorm).
__global__ void __launch_bounds__(256)
norm_bwd_means_kernel(const float* __restrict__ s, const float* __restrict__ gamma, float* __restrict__ M,
                      float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int groups, float rows, int cpg_arg) {
  // cpg_arg > 0: `groups` groups of cpg_arg channels; channels from groups * cpg_arg on are alignment padding (all zero tensors):
  // M = 0 there and gamma (groups * cpg_arg entries) is not read
  const int cpg_ = groups > 0 ? (cpg_arg > 0 ? cpg_arg : C / groups) : 0;
  const int C_real = groups > 0 ? groups * cpg_ : C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a0 = 0.f, a1 = 0.f;
    for (int n = 0; n < N; ++n) { a0 += s[(n * 2 + 0) * C + c]; a1 += s[(n * 2 + 1) * C + c]; }
    if (dbeta) dbeta[c] = a0;
    if (dgamma) dgamma[c] = a1;
    if (c >= C_real)
      for (int n = 0; n < N; ++n) { M[(n * 2 + 0) * C + c] = 0.f; M[(n * 2 + 1) * C + c] = 0.f; }
    if (groups == 0) {
      const float gm = gamma ? gamma[c] : 1.0f;
      const float inv = 1.0f / ((float)N * rows);
      for (int n = 0; n < N; ++n) { M[(n * 2 + 0) * C + c] = a0 * gm * inv; M[(n * 2 + 1) * C + c] = a1 * gm * inv; }
    }
  }
  if (groups > 0) {
    const int cpg = cpg_;
    const float inv = 1.0f / (rows * (float)cpg);
    for (int i = threadIdx.x; i < N * 2 * groups; i += 256) {
      const int g = i % groups, nk = i / groups;
      float a = 0.f;
      for (int j = 0; j < cpg; ++j) { const int c = g * cpg + j; a += s[nk * C + c] * (gamma ? gamma[c] : 1.0f); }
      a *= inv;
      for (int j = 0; j < cpg; ++j) M[nk * C + g * cpg + j] = a;
    }
  }
}

// BatchNorm running buffers from the batch statistics (mean, rstd) of one forward: unbiased variance, momentum blend
__global__ void __launch_bounds__(256)
bn_update_running_kernel(const float* __restrict__ mr, float* __restrict__ rmean, float* __restrict__ rvar, int C, float n,
                         float eps, float momentum) {
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mean = mr[c], rstd = mr[C + c];
    float var = 1.0f / (rstd * rstd) - eps;
    var = (var > 0.f ? var : 0.f) * (n / (n > 1.f ? n - 1.f : 1.f));
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var;
  }
}

// BatchNorm3d in train() mode, everything after the statistics pass in ONE launch: per channel the fixed-order reduction of
// the slot partials of ALL samples (same tree as norm_finalize_groups_kernel with groups = C on the flattened slots), the affine
// (a, b) and (mean, rstd) written for every sample of the batch, the running_mean / running_var blend (same expressions as
// bn_update_running_kernel) and num_batches_tracked += 1.  Replaces 5 launches per NormAct forward (finalize, two expand copies,
// the counter increment, the running update): RSUNet and the MONAI U-Net are launch bound at their patch sizes.
__global__ void __launch_bounds__(256)
bn_train_finalize_kernel(const float* __restrict__ stats, int slots, float count, const float* __restrict__ gamma,
                         const float* __restrict__ beta, float eps, float momentum, float* __restrict__ rmean,
                         float* __restrict__ rvar, long long* __restrict__ nbt, float* __restrict__ ab, float* __restrict__ mr,
                         int N, int C, int C_real) {
  __shared__ float red[2][256];
  const int c = blockIdx.x;
  if (c >= C_real) {        // alignment padding (all-zero channels): affine (0, 0), no parameters or running buffers behind them
    if (threadIdx.x == 0)
      for (int n = 0; n < N; ++n) {
        ab[((long)n * 2 + 0) * C + c] = 0.f; ab[((long)n * 2 + 1) * C + c] = 0.f;
        mr[((long)n * 2 + 0) * C + c] = 0.f; mr[((long)n * 2 + 1) * C + c] = 0.f;
      }
    return;
  }
  float a1 = 0.f, a2 = 0.f;
  for (long s = threadIdx.x; s < slots; s += blockDim.x) {
    a1 += stats[(s * 2 + 0) * C + c];
    a2 += stats[(s * 2 + 1) * C + c];
  }
  red[0][threadIdx.x] = a1;
  red[1][threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t1 = 0.f, t2 = 0.f;
    for (int i = 0; i < 256; ++i) { t1 += red[0][i]; t2 += red[1][i]; }
    const float mean = t1 / count;
    const float var = fmaxf(t2 / count - mean * mean, 0.f);
    float rstd = rsqrtf(var + eps);
    rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);
    const float a = (gamma ? gamma[c] : 1.f) * rstd;
    const float b = (beta ? beta[c] : 0.f) - mean * a;
    for (int n = 0; n < N; ++n) {
      ab[((long)n * 2 + 0) * C + c] = a;
      ab[((long)n * 2 + 1) * C + c] = b;
      mr[((long)n * 2 + 0) * C + c] = mean;
      mr[((long)n * 2 + 1) * C + c] = rstd;
    }
    if (rmean) {
      float v = 1.0f / (rstd * rstd) - eps;
      v = (v > 0.f ? v : 0.f) * (count / (count > 1.f ? count - 1.f : 1.f));
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * v;
    }
    if (nbt && c == 0) *nbt += 1;
  }
}

}  // namespace pytc

using namespace pytc;

#define RS_DISPATCH(dtype, BF, F32, what)                                   \
  if (dtype == PYTC_BF16) { BF; } else if (dtype == PYTC_F32) { F32; } else { \
    set_error(what ": bad dtype %d", dtype); return PYTC_ERR_INVALID; }

// launch plan shared by the workspace query and the launch
struct CwPlan { bool mfma, rag_o, rag_k; int mt, nt, kp, tiles, slots; long per_slot; };
// one input and one output channel, 3^3 taps (the last conv of a single-class U-Net): a 27-bin correlation of two scalar fields
static bool cw_scalar(int C_in, int C_out, const int32_t* k, int dtype) {
  return C_in == 1 && C_out == 1 && k[0] == 3 && k[1] == 3 && k[2] == 3 && (dtype == PYTC_BF16 || dtype == PYTC_F32) &&
         tuning_get("conv_wgrad_scalar", 1) != 0;
}

static CwPlan cw_plan(int N, int D, int H, int W, int C_in, int C_out, const int32_t* k, int dtype) {
  CwPlan p{};
  const long rows_total = (long)N * D * H * W;
  if (cw_scalar(C_in, C_out, k, dtype)) {
    const long s = (rows_total + 4095) / 4096;          // 16 voxels per thread
    p.slots = (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
    p.per_slot = (rows_total + p.slots - 1) / p.slots;
    return p;
  }
  p.rag_o = C_out < 16;
  p.rag_k = C_in < 16;
  p.kp = k[1];
  // rows must be 16-byte aligned (C % 8 == 0); a channel count that is not a multiple of the tile width leaves the last tile
  // partly filled (24 = 16 + 8, 40 = 16 + 16 + 8: RSUNet's stock widths [18, 36, ...] padded to 8 by the model)
  p.mfma = dtype == PYTC_BF16 && k[1] == k[2] && (k[1] == 3 || k[1] == 1) && (p.rag_o || C_out % 8 == 0) &&
           (p.rag_k || C_in % 8 == 0) && (tuning_get("conv_wgrad_mfma", 1) != 0);
  if (p.mfma) {
    p.mt = (!p.rag_o && C_out % 32 == 0) ? 2 : 1;
    p.nt = (!p.rag_k && C_in % 32 == 0) ? 2 : 1;
    p.tiles = (p.rag_o ? 1 : (C_out + p.mt * 16 - 1) / (p.mt * 16)) * (p.rag_k ? 1 : (C_in + p.nt * 16 - 1) / (p.nt * 16));
    const long units = (long)N * D * H * ((W + 31) / 32);
    long s = 1536 / ((long)p.tiles * k[0]);            // ~1536 workgroups over the launch (6 per CU)
    if (s > units / 16) s = units / 16;                // >= 4 consecutive units per wave: the line ring needs a run to pay off
    s = (s / 8) * 8;
    p.slots = (int)(s < 8 ? 8 : (s > 512 ? 512 : s));
    p.per_slot = (units + p.slots - 1) / p.slots;
  } else {
    const long s = rows_total / 8192;
    p.slots = (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
    p.per_slot = (rows_total + p.slots - 1) / p.slots;
  }
  return p;
}

// dW[t] = sum_v dy[v] * a[v + t - 1]: every thread walks its share of a slot's voxels with 27 accumulators (inputs loaded branch-free:
// clamped address, zeroed outside the volume), the workgroup sums them in a fixed order (xor shuffles, then the four waves in LDS) ->
// ws[slot][27]; reduce_slots2 finishes.  The MFMA kernel gave this 1 x 1 problem one lane of a 16 x 16 tile: 210 us at 2 x 24 x 256 x 256.
template <typename T>
__global__ void __launch_bounds__(256)
conv3d_wgrad_scalar_kernel(const T* __restrict__ a, const T* __restrict__ dy, float* __restrict__ ws, int N, int D, int H, int W,
                           long per_slot, long rows_total) {
  __shared__ float red[4][27];
  const long v0 = (long)blockIdx.x * per_slot;
  const long v1 = v0 + per_slot < rows_total ? v0 + per_slot : rows_total;
  const long rps = (long)D * H * W;
  float acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  for (long v = v0 + threadIdx.x; v < v1; v += 256) {
    const int n = (int)(v / rps);
    const long row = v - (long)n * rps;
    const int x = (int)(row % W);
    const long tq = row / W;
    const int y = (int)(tq % H), z = (int)(tq / H);
    const float g = to_f32<T>(dy[v]);
    const T* an = a + (long)n * rps;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int dyy = 0; dyy < 3; ++dyy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int zz = z + dz - 1, yy = y + dyy - 1, xx = x + dx - 1;
          const bool ok = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
          const float t = to_f32<T>(an[((long)min(max(zz, 0), D - 1) * H + min(max(yy, 0), H - 1)) * W + min(max(xx, 0), W - 1)]);
          acc[(dz * 3 + dyy) * 3 + dx] = fmaf(g, ok ? t : 0.f, acc[(dz * 3 + dyy) * 3 + dx]);
        }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    float s = acc[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[wave][t] = s;
  }
  __syncthreads();
  if (threadIdx.x < 27) ws[(long)blockIdx.x * 27 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

extern "C" int64_t pytc_conv3d_wgrad_ws_elems(int N, int D, int H, int W, int C_in, int C_out, const int32_t* kernel,
                                              int dtype) {
  if (!kernel || N < 1) return -1;
  const CwPlan p = cw_plan(N, D, H, W, C_in, C_out, kernel, dtype);
  return (int64_t)p.slots * kernel[0] * kernel[1] * kernel[2] * C_out * C_in;
}

extern "C" int pytc_conv3d_wgrad(const void* a, const void* dy, float* dW, float* workspace, int N, int D, int H, int W,
                                 int C_in, int C_out, const int32_t* kernel, int dtype, void* stream) {
  PYTC_REQUIRE(a && dy && dW && workspace && kernel && N >= 1, "conv3d_wgrad: bad arguments");
  CwGeom g{D, H, W, kernel[0], kernel[1], kernel[2]};
  PYTC_REQUIRE(g.kd % 2 == 1 && g.kh % 2 == 1 && g.kw % 2 == 1, "conv3d_wgrad: odd kernel sizes only ('same' padding)");
  const long rows_total = (long)N * D * H * W;
  const CwPlan p = cw_plan(N, D, H, W, C_in, C_out, kernel, dtype);
  const int slots = p.slots;
  const int taps = g.kd * g.kh * g.kw;
  hipStream_t s = (hipStream_t)stream;
  if (cw_scalar(C_in, C_out, kernel, dtype)) {
    RS_DISPATCH(dtype,
                hipLaunchKernelGGL(conv3d_wgrad_scalar_kernel<bf16_t>, dim3(slots), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)dy, workspace, N, D, H, W, p.per_slot, rows_total),
                hipLaunchKernelGGL(conv3d_wgrad_scalar_kernel<float>, dim3(slots), dim3(256), 0, s, (const float*)a, (const float*)dy, workspace, N, D, H, W, p.per_slot, rows_total),
                "conv3d_wgrad")
  } else if (p.mfma) {
    const long blocks = (long)((slots + 7) / 8) * 8 * p.tiles * g.kd;
    const bf16_t* ap = (const bf16_t*)a;
    const bf16_t* dp = (const bf16_t*)dy;
#define CW_LAUNCH(MT, NT, KP, RO, RK) hipLaunchKernelGGL((conv3d_wgrad_mfma_kernel<MT, NT, KP, RO, RK>), dim3((unsigned)blocks), dim3(256), 0, s, ap, dp, workspace, N, g, C_in, C_out, p.per_slot, slots, p.tiles)
#define CW_SHAPES(KP)                                                         \
    if (p.rag_o && p.rag_k) CW_LAUNCH(1, 1, KP, true, true);                  \
    else if (p.rag_o && p.nt == 2) CW_LAUNCH(1, 2, KP, true, false);          \
    else if (p.rag_o) CW_LAUNCH(1, 1, KP, true, false);                       \
    else if (p.rag_k && p.mt == 2) CW_LAUNCH(2, 1, KP, false, true);          \
    else if (p.rag_k) CW_LAUNCH(1, 1, KP, false, true);                       \
    else if (p.mt == 2 && p.nt == 2) CW_LAUNCH(2, 2, KP, false, false);       \
    else if (p.mt == 2) CW_LAUNCH(2, 1, KP, false, false);                    \
    else if (p.nt == 2) CW_LAUNCH(1, 2, KP, false, false);                    \
    else CW_LAUNCH(1, 1, KP, false, false);
    if (p.kp == 3) { CW_SHAPES(3) } else { CW_SHAPES(1) }
#undef CW_SHAPES
#undef CW_LAUNCH
  } else {
    const long rps = p.per_slot;
    dim3 grid(slots, ((C_out + CW_TO - 1) / CW_TO) * ((C_in + CW_TK - 1) / CW_TK), taps), block(256);
    RS_DISPATCH(dtype,
                hipLaunchKernelGGL(conv3d_wgrad_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)a, (const bf16_t*)dy, workspace, rows_total, g, C_in, C_out, rps, slots),
                hipLaunchKernelGGL(conv3d_wgrad_kernel<float>, grid, block, 0, s, (const float*)a, (const float*)dy, workspace, rows_total, g, C_in, C_out, rps, slots),
                "conv3d_wgrad")
  }
  const long nW = (long)taps * C_out * C_in;
  hipLaunchKernelGGL(reduce_slots2_kernel, dim3(ceil_div(nW, 16)), dim3(256), 0, s, workspace, dW, nW, slots);
  PYTC_LAUNCH_CHECK("conv3d_wgrad");
  return PYTC_OK;
}

extern "C" int pytc_act_bwd(const void* da, const void* x, const float* ab, void* dt, void* dp, int N, int64_t rows, int C,
                            int act, float prm, int dtype, void* stream) {
  PYTC_REQUIRE(da && x && dt && N >= 1 && rows >= 1 && C >= 1, "act_bwd: bad arguments");
  const long total = (long)N * rows * C;
  hipStream_t s = (hipStream_t)stream;
  {
    const int vec = dtype == PYTC_BF16 ? 8 : 4;
    if ((dtype == PYTC_BF16 || dtype == PYTC_F32) && C % vec == 0 && tuning_get("elementwise_vec", 1)) {
      const long chunks = total / vec;
      if (dtype == PYTC_BF16)
        hipLaunchKernelGGL(act_bwd_vec_kernel<bf16_t>, dim3(grid_for(chunks)), dim3(256), 0, s, (const bf16_t*)da, (const bf16_t*)x, ab, (bf16_t*)dt, (bf16_t*)dp, (long)rows, C, chunks, act, prm);
      else
        hipLaunchKernelGGL(act_bwd_vec_kernel<float>, dim3(grid_for(chunks)), dim3(256), 0, s, (const float*)da, (const float*)x, ab, (float*)dt, (float*)dp, (long)rows, C, chunks, act, prm);
      PYTC_LAUNCH_CHECK("act_bwd");
      return PYTC_OK;
    }
  }
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)da, (const bf16_t*)x, ab, (bf16_t*)dt, (bf16_t*)dp, (long)rows, C, total, act, prm),
              hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)da, (const float*)x, ab, (float*)dt, (float*)dp, (long)rows, C, total, act, prm),
              "act_bwd")
  PYTC_LAUNCH_CHECK("act_bwd");
  return PYTC_OK;
}

extern "C" int pytc_norm_bwd_apply_general(const void* d, const void* x, const float* mean_rstd, const float* gamma,
                                           const float* M, void* dx, int N, int64_t rows, int C, int dtype, void* stream) {
  PYTC_REQUIRE(d && x && mean_rstd && M && dx, "norm_bwd_apply_general: null pointer");
  const long total = (long)N * rows * C;
  hipStream_t s = (hipStream_t)stream;
  {
    const int vec = dtype == PYTC_BF16 ? 8 : 4;
    if ((dtype == PYTC_BF16 || dtype == PYTC_F32) && C % vec == 0 && tuning_get("elementwise_vec", 1)) {
      const long chunks = total / vec;
      if (dtype == PYTC_BF16)
        hipLaunchKernelGGL(norm_bwd_apply_general_vec_kernel<bf16_t>, dim3(grid_for(chunks)), dim3(256), 0, s, (const bf16_t*)d, (const bf16_t*)x, mean_rstd, gamma, M, (bf16_t*)dx, (long)rows, C, chunks);
      else
        hipLaunchKernelGGL(norm_bwd_apply_general_vec_kernel<float>, dim3(grid_for(chunks)), dim3(256), 0, s, (const float*)d, (const float*)x, mean_rstd, gamma, M, (float*)dx, (long)rows, C, chunks);
      PYTC_LAUNCH_CHECK("norm_bwd_apply_general");
      return PYTC_OK;
    }
  }
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(norm_bwd_apply_general_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)d, (const bf16_t*)x, mean_rstd, gamma, M, (bf16_t*)dx, (long)rows, C, total),
              hipLaunchKernelGGL(norm_bwd_apply_general_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)d, (const float*)x, mean_rstd, gamma, M, (float*)dx, (long)rows, C, total),
              "norm_bwd_apply_general")
  PYTC_LAUNCH_CHECK("norm_bwd_apply_general");
  return PYTC_OK;
}

/* act_bwd + norm_bwd_stats without the intermediate dt: s_out [N][2][C] = (sum dt, sum dt * xhat), dt = da * act'(a*x + b) rounded
   to the storage type; p_out (nullable, PReLU) [N][C] = sum da * min(a*x + b, 0).  stats_ws: pytc_norm_bwd_ws_elems floats, p_ws:
   half of that.  C a multiple of 8 (bf16) / 4 (fp32). */
extern "C" int pytc_act_norm_bwd_stats(const void* da, const void* x, const float* ab, const float* mean_rstd, float* stats_ws,
                                       float* s_out, float* p_ws, float* p_out, int N, int64_t rows, int C, int act, float prm,
                                       int dtype, void* stream) {
  PYTC_REQUIRE(da && x && mean_rstd && stats_ws && s_out && N >= 1 && rows >= 1, "act_norm_bwd_stats: bad arguments");
  PYTC_REQUIRE((p_ws == nullptr) == (p_out == nullptr), "act_norm_bwd_stats: p_ws and p_out come as a pair");
  PYTC_REQUIRE((dtype == PYTC_BF16 && C % 8 == 0) || (dtype == PYTC_F32 && C % 4 == 0), "act_norm_bwd_stats: C %% (16 bytes) != 0");
  const int slots = act_norm_slots(rows, C, dtype == PYTC_BF16 ? 8 : 4);      // <= colstats_slots(rows): the workspace bound
  const long rps = (rows + slots - 1) / slots;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(slots, N), block(256);
  if (dtype == PYTC_BF16) {
    if (p_ws) hipLaunchKernelGGL((act_norm_bwd_stats_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)da, (const bf16_t*)x, ab, mean_rstd, stats_ws, p_ws, (long)rows, C, slots, rps, act, prm);
    else hipLaunchKernelGGL((act_norm_bwd_stats_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)da, (const bf16_t*)x, ab, mean_rstd, stats_ws, p_ws, (long)rows, C, slots, rps, act, prm);
  } else {
    if (p_ws) hipLaunchKernelGGL((act_norm_bwd_stats_kernel<float, true>), grid, block, 0, s, (const float*)da, (const float*)x, ab, mean_rstd, stats_ws, p_ws, (long)rows, C, slots, rps, act, prm);
    else hipLaunchKernelGGL((act_norm_bwd_stats_kernel<float, false>), grid, block, 0, s, (const float*)da, (const float*)x, ab, mean_rstd, stats_ws, p_ws, (long)rows, C, slots, rps, act, prm);
  }
  hipLaunchKernelGGL(reduce_slots_batched_rs_kernel, dim3(ceil_div(2L * C, 16), N), dim3(256), 0, s, stats_ws, s_out, 2L * C, slots);
  if (p_ws) hipLaunchKernelGGL(reduce_slots_batched_rs_kernel, dim3(ceil_div((long)C, 16), N), dim3(256), 0, s, p_ws, p_out, (long)C, slots);
  PYTC_LAUNCH_CHECK("act_norm_bwd_stats");
  return PYTC_OK;
}

/* act_bwd + norm_bwd_apply_general without the intermediate dt: dx = rstd * (gamma * dt - M1 - xhat * M2) */
static int act_norm_bwd_apply_impl(const void* da, const void* x, const float* ab, const float* mean_rstd, const float* gamma, int C_gamma,
                                   const float* M, void* dx, int N, int64_t rows, int C, int act, float prm, int dtype, void* stream);

extern "C" int pytc_act_norm_bwd_apply(const void* da, const void* x, const float* ab, const float* mean_rstd, const float* gamma,
                                       const float* M, void* dx, int N, int64_t rows, int C, int act, float prm, int dtype,
                                       void* stream) {
  return act_norm_bwd_apply_impl(da, x, ab, mean_rstd, gamma, C, M, dx, N, rows, C, act, prm, dtype, stream);
}

/* gamma holds C_gamma <= C entries (the norm's own channels next to channel-padded activations); the padding takes gamma = 0 */
extern "C" int pytc_act_norm_bwd_apply_cg(const void* da, const void* x, const float* ab, const float* mean_rstd, const float* gamma,
                                          int C_gamma, const float* M, void* dx, int N, int64_t rows, int C, int act, float prm, int dtype,
                                          void* stream) {
  PYTC_REQUIRE(C_gamma >= 0 && C_gamma <= C, "act_norm_bwd_apply: C_gamma = %d outside [0, %d]", C_gamma, C);
  return act_norm_bwd_apply_impl(da, x, ab, mean_rstd, gamma, C_gamma, M, dx, N, rows, C, act, prm, dtype, stream);
}

static int act_norm_bwd_apply_impl(const void* da, const void* x, const float* ab, const float* mean_rstd, const float* gamma, int C_gamma,
                                   const float* M, void* dx, int N, int64_t rows, int C, int act, float prm, int dtype, void* stream) {
  PYTC_REQUIRE(da && x && mean_rstd && M && dx, "act_norm_bwd_apply: null pointer");
  PYTC_REQUIRE((dtype == PYTC_BF16 && C % 8 == 0) || (dtype == PYTC_F32 && C % 4 == 0), "act_norm_bwd_apply: C %% (16 bytes) != 0");
  const int slots = act_norm_slots(rows, C, dtype == PYTC_BF16 ? 8 : 4);
  const long rps = (rows + slots - 1) / slots;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(slots, N), block(256);
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(act_norm_bwd_apply_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)da, (const bf16_t*)x, ab, mean_rstd, gamma, M, (bf16_t*)dx, (long)rows, C, rps, act, prm, C_gamma);
  else
    hipLaunchKernelGGL(act_norm_bwd_apply_kernel<float>, grid, block, 0, s, (const float*)da, (const float*)x, ab, mean_rstd, gamma, M, (float*)dx, (long)rows, C, rps, act, prm, C_gamma);
  PYTC_LAUNCH_CHECK("act_norm_bwd_apply");
  return PYTC_OK;
}

extern "C" int pytc_maxpool3d_bwd(const void* x, const void* dy, void* dx, int N, int D, int H, int W, int C, int fz, int fy,
                                  int fx, int dtype, void* stream) {
  PYTC_REQUIRE(x && dy && dx && fz >= 1 && fy >= 1 && fx >= 1, "maxpool3d_bwd: bad arguments");
  const int Do = D / fz, Ho = H / fy, Wo = W / fx;
  const long total = (long)N * Do * Ho * Wo * C;
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)N * D * H * W * C * (dtype == PYTC_BF16 ? 2 : 4);
  if (hipMemsetAsync(dx, 0, bytes, s) != hipSuccess) { set_error("maxpool3d_bwd: memset failed"); return PYTC_ERR_HIP; }
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(maxpool3d_bwd_kernel<bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, D, H, W, C, fz, fy, fx, Do, Ho, Wo, total),
              hipLaunchKernelGGL(maxpool3d_bwd_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)dx, D, H, W, C, fz, fy, fx, Do, Ho, Wo, total),
              "maxpool3d_bwd")
  PYTC_LAUNCH_CHECK("maxpool3d_bwd");
  return PYTC_OK;
}

extern "C" int pytc_dwconv3d_generic_fwd(const void* x, void* y, const float* w, int N, int D, int H, int W, int C,
                                         const int32_t* kernel, const int32_t* stride, const int32_t* pad,
                                         const int32_t* out_dims, int dtype, void* stream) {
  PYTC_REQUIRE(x && y && w && kernel && stride && pad && out_dims, "dwconv3d_generic: null pointer");
  DwGen g{D, H, W, C, kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
          out_dims[0], out_dims[1], out_dims[2]};
  const long total = (long)N * g.Do * g.Ho * g.Wo * C;
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH(dtype,
              hipLaunchKernelGGL(dwconv3d_generic_kernel<bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, w, g, total),
              hipLaunchKernelGGL(dwconv3d_generic_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)x, (float*)y, w, g, total),
              "dwconv3d_generic")
  PYTC_LAUNCH_CHECK("dwconv3d_generic");
  return PYTC_OK;
}

extern "C" int pytc_norm_bwd_means_cpg(const float* s, const float* gamma, float* M, float* dgamma, float* dbeta, int N, int C,
                                       int groups, int cpg, float rows, void* stream) {
  PYTC_REQUIRE(s && M && N >= 1 && C >= 1 && rows >= 1.f, "norm_bwd_means: bad arguments");
  PYTC_REQUIRE(groups == 0 || (groups >= 1 && (cpg > 0 ? groups * cpg <= C : C % groups == 0)),
               "norm_bwd_means: C must be divisible by groups (or groups * cpg <= C with explicit channels per group)");
  hipLaunchKernelGGL(norm_bwd_means_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s, gamma, M, dgamma, dbeta, N, C, groups, rows,
                     cpg);
  PYTC_LAUNCH_CHECK("norm_bwd_means");
  return PYTC_OK;
}

extern "C" int pytc_norm_bwd_means(const float* s, const float* gamma, float* M, float* dgamma, float* dbeta, int N, int C,
                                   int groups, float rows, void* stream) {
  return pytc_norm_bwd_means_cpg(s, gamma, M, dgamma, dbeta, N, C, groups, 0, rows, stream);
}

extern "C" int pytc_bn_train_finalize_cpad(const float* stats, int slots_total, float count, const float* gamma, const float* beta,
                                           float eps, float momentum, float* running_mean, float* running_var,
                                           int64_t* num_batches_tracked, float* ab, float* mean_rstd, int N, int C, int C_real,
                                           void* stream) {
  PYTC_REQUIRE(stats && ab && mean_rstd && slots_total >= 1 && count >= 1.f && N >= 1 && C >= 1 && C_real >= 1 && C_real <= C,
               "bn_train_finalize: bad arguments");
  PYTC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_finalize: running buffers come as a pair");
  hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, slots_total, count, gamma, beta,
                     eps, momentum, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), ab, mean_rstd, N,
                     C, C_real);
  PYTC_LAUNCH_CHECK("bn_train_finalize");
  return PYTC_OK;
}

extern "C" int pytc_bn_train_finalize(const float* stats, int slots_total, float count, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var,
                                      int64_t* num_batches_tracked, float* ab, float* mean_rstd, int N, int C, void* stream) {
  return pytc_bn_train_finalize_cpad(stats, slots_total, count, gamma, beta, eps, momentum, running_mean, running_var,
                                     num_batches_tracked, ab, mean_rstd, N, C, C, stream);
}

extern "C" int pytc_bn_update_running(const float* mean_rstd, float* running_mean, float* running_var, int C, float count,
                                      float eps, float momentum, void* stream) {
  PYTC_REQUIRE(mean_rstd && running_mean && running_var && C >= 1 && count >= 1.f, "bn_update_running: bad arguments");
  hipLaunchKernelGGL(bn_update_running_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean_rstd, running_mean, running_var, C,
                     count, eps, momentum);
  PYTC_LAUNCH_CHECK("bn_update_running");
  return PYTC_OK;
}
