// Dense Conv3d (stride 1, "same" zero padding, odd kernel per axis, no groups) as an implicit GEMM on MFMA,
// in the transposed form of pw_kernels.hip:  Y^T[o][v] = sum_{tap,c} W[o][tap][c] * f(X)^T[c][v + tap]
// where f is the optional fused pre-activation  f(x) = act(a[n][c]*x + b[n][c])  (RSUNet's NormAct ahead of
// every conv, rsunet.py:87-118,145-198) -- zero padding applies to f(X), so out-of-volume taps contribute 0.
// A lane's B fragment for one (tap, k-group) is 8 (bf16) / 4 (fp32) consecutive channels of ONE neighbouring
// voxel: a 16-byte NDHWC load; neighbouring voxels' loads hit L1/L2 (27x reuse).
#include "pw_common.h"

namespace pytc {

struct ConvParams {
  const void* x;
  const void* wp;       // [mtile][tap][kgroup][lane][EPL]
  const float* bias;
  const float* ab;      // [N][2][C_in] or NULL
  EpiParams e;
  int N, D, H, W, C_in, C_out, KG, MTt;
  int kd, kh, kw;
  int act_in;           // PYTC_ACT_* of the fused pre-activation (NONE = affine only)
  float act_param;      // leaky slope / prelu weight / elu alpha
  int act_out;
  // LDS-tiled form only (round 6): low-side zero padding per axis (a 'same' conv: k / 2) and the output map -- om = 1: the voxel
  // (z, y, x) of the D x H x W grid the kernel walks is stored at (2z + oz, 2y + oy, 2x + ox) of a Do x Ho x Wo tensor (one PHASE of a
  // stride-2 transposed conv: csrc/conv3d_kernels.hip convT phase form)
  int pd, ph, pw;
  int om, oz, oy, ox, Ho, Wo;
};

__device__ __forceinline__ float pre_act(float v, int act, float prm) {
  switch (act) {
    case PYTC_ACT_RELU: return fmaxf(v, 0.f);
    case PYTC_ACT_LEAKY: return v > 0.f ? v : v * prm;
    case PYTC_ACT_ELU: return v > 0.f ? v : prm * (__expf(v) - 1.0f);
    default: return v;
  }
}

template <typename TI, typename TW, typename TO, int MT, int NT>
__global__ void __launch_bounds__(256)
conv3d_kernel(ConvParams p) {
  typedef Mma<TW> M;
  constexpr int EPL = M::EPL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.z;
  const int mt0 = blockIdx.y * MT;
  const long rps = (long)p.D * p.H * p.W;
  const long row0 = ((long)blockIdx.x * 4 + wave) * (NT * 16);
  if (row0 >= rps) return;
  const int r = lane & 15, kb = lane >> 4;

  long orow[NT];
  int vz[NT], vy[NT], vx[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    long o = row0 + nt * 16 + r;
    orow[nt] = o;
    long oc = o < rps ? o : rps - 1;
    vx[nt] = (int)(oc % p.W);
    long t = oc / p.W;
    vy[nt] = (int)(t % p.H);
    vz[nt] = (int)(t / p.H);
  }
  const TI* xn = reinterpret_cast<const TI*>(p.x) + (long)n * rps * p.C_in;
  const bool vec_ok = (p.C_in % EPL) == 0;
  const bool affine = p.ab != nullptr;

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const typename M::frag_t* wp = reinterpret_cast<const typename M::frag_t*>(p.wp);
  const int ntap = p.kd * p.kh * p.kw;
  const int pd = p.kd / 2, ph = p.kh / 2, pw = p.kw / 2;
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
      const int dx = tap % p.kw - pw;
      const int tt = tap / p.kw;
      const int dy = tt % p.kh - ph;
      const int dz = tt / p.kh - pd;
      typename M::frag_t bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int z = vz[nt] + dz, y = vy[nt] + dy, x = vx[nt] + dx;
        float v[EPL];
        if (z >= 0 && z < p.D && y >= 0 && y < p.H && x >= 0 && x < p.W) {
          load_row_frag<TI, EPL>(xn + (((long)z * p.H + y) * p.W + x) * p.C_in, k0, p.C_in, vec_ok, v);
          if (affine) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) v[j] = fmaf(v[j], av[j], bv[j]);
          }
          if (p.act_in != PYTC_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) v[j] = (k0 + j < p.C_in) ? pre_act(v[j], p.act_in, p.act_param) : 0.f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < EPL; ++j) v[j] = 0.f;
        }
        bf[nt] = M::from_floats(v);
      }
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
      for (int rr = 0; rr < 4; ++rr) v[rr] = apply_act(acc[mt][nt][rr] + bo[rr], p.act_out);
      finish_and_store<TO, 4>(v, p.e, n, orow[nt], o0);
    }
  }
}

// ---- thin-input form (C_in <= 4: the network's first conv) --------------------------------------------------------------
// HBM bound (reads C_in, writes C_out channels per voxel) with 27*C_in*C_out FMAs per voxel: one lane = one output voxel
// x 16 output channels on the VALU, weights as fp32 in LDS (broadcast reads), neighbour voxels from L1/L2.  Reads the
// same packed weight image as the MFMA kernel above.
template <typename TI, typename TW, typename TO>
__global__ void __launch_bounds__(256)
conv3d_thin_in_kernel(ConvParams p) {
  typedef Mma<TW> M;
  constexpr int EPL = M::EPL;
  extern __shared__ __attribute__((aligned(16))) float wl[];            // [tap][ci][16]
  const int n = blockIdx.z, oc0 = blockIdx.y * 16;
  const int ntap = p.kd * p.kh * p.kw;
  const TW* wp = reinterpret_cast<const TW*>(p.wp);
  for (int i = threadIdx.x; i < ntap * p.C_in * 16; i += 256) {
    const int o = i % 16, ci = (i / 16) % p.C_in, tap = i / (16 * p.C_in);
    const long q = ((((long)blockIdx.y * ntap + tap) * p.KG + ci / M::KSTEP) * 64 + ((ci % M::KSTEP) / EPL) * 16 + o) * EPL + ci % EPL;
    wl[i] = to_f32<TW>(wp[q]);
  }
  __syncthreads();
  const long rps = (long)p.D * p.H * p.W;
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rps) return;
  const int vx = (int)(row % p.W);
  const long tq = row / p.W;
  const int vy = (int)(tq % p.H), vz = (int)(tq / p.H);
  const TI* xn = reinterpret_cast<const TI*>(p.x) + (long)n * rps * p.C_in;
  const int pd = p.kd / 2, ph = p.kh / 2, pw = p.kw / 2;
  float acc[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc[o] = 0.f;
  int tap = 0;
  for (int dz = -pd; dz <= pd; ++dz)
    for (int dy = -ph; dy <= ph; ++dy)
      for (int dx = -pw; dx <= pw; ++dx, ++tap) {
        const int z = vz + dz, y = vy + dy, x = vx + dx;
        if (z < 0 || z >= p.D || y < 0 || y >= p.H || x < 0 || x >= p.W) continue;      // zero padding of f(X)
        const TI* xv = xn + (((long)z * p.H + y) * p.W + x) * p.C_in;
        for (int ci = 0; ci < p.C_in; ++ci) {
          float v = to_f32<TI>(xv[ci]);
          if (p.ab) v = fmaf(v, p.ab[((long)n * 2 + 0) * p.C_in + ci], p.ab[((long)n * 2 + 1) * p.C_in + ci]);
          if (p.act_in != PYTC_ACT_NONE) v = pre_act(v, p.act_in, p.act_param);
          v = to_f32<TW>(from_f32<TW>(v));                 // the MFMA kernels consume the operand rounded to TW
          const float4* wq = reinterpret_cast<const float4*>(wl + (tap * p.C_in + ci) * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 w4 = wq[q];
            acc[q * 4 + 0] = fmaf(v, w4.x, acc[q * 4 + 0]); acc[q * 4 + 1] = fmaf(v, w4.y, acc[q * 4 + 1]);
            acc[q * 4 + 2] = fmaf(v, w4.z, acc[q * 4 + 2]); acc[q * 4 + 3] = fmaf(v, w4.w, acc[q * 4 + 3]);
          }
        }
      }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o0 = oc0 + q * 4;
    if (o0 >= p.C_out) continue;
    float v[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      v[rr] = apply_act(acc[q * 4 + rr] + ((p.bias && o0 + rr < p.C_out) ? p.bias[o0 + rr] : 0.f), p.act_out);
    finish_and_store<TO, 4>(v, p.e, n, row, o0);
  }
}

// ---- one input channel, 3 x 3 x 3, stride 1 or 2 (round 6) --------------------------------------------------------------------------------
// The first conv of a U-Net (1 -> 32 / 64, stride 2 in the MONAI-style network) and the 1 -> 1 conv of its last residual unit are
// stencils on a SCALAR field: 27 input loads and 27 x C_out FMAs per output voxel, HBM bound on the output.  On the MFMA gather kernel
// the single channel is one of 32 K slots (conv3d_s_fwd[1 -> 32] 156 us, [1 -> 64] 258 us at 2 x 24 x 256 x 256), on the thin-input
// kernel above a 1 -> 1 conv pays 16 output slots and a branch per tap (119 us).  Here: one thread = one output voxel x up to OCT
// output channels, the 27 inputs loaded branch-free (clamped address, zeroed when outside) before any arithmetic, weights (the bf16 /
// fp32 image the MFMA kernels read, tap-major layout) as fp32 in LDS with wave-uniform broadcast reads, 16-byte stores.
template <typename T, int OCT>
__global__ void __launch_bounds__(256)
conv3d_c1_stencil_kernel(const T* __restrict__ x, const T* __restrict__ wp, const float* __restrict__ bias, T* __restrict__ y, int N,
                         int Do, int Ho, int Wo, int Di, int Hi, int Wi, int C_out, int stride) {
  constexpr int EPL = Mma<T>::EPL;
  __shared__ float wl[27 * OCT];
  const int oc0 = blockIdx.y * OCT;
  for (int i = threadIdx.x; i < 27 * OCT; i += 256) {
    const int o = oc0 + i % OCT, tap = i / OCT;
    // [mtile][tap][kgroup = 0][lane = (kb = 0, r = o % 16)][j = 0]: input channel 0 of output channel o
    wl[i] = o < C_out ? to_f32<T>(wp[(((long)(o >> 4) * 27 + tap) * 64 + (o & 15)) * EPL]) : 0.f;
  }
  __syncthreads();
  const long rps = (long)Do * Ho * Wo;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= rps * N) return;
  const int n = (int)(v / rps);
  const long row = v - (long)n * rps;
  const int vx = (int)(row % Wo);
  const long tq = row / Wo;
  const int vy = (int)(tq % Ho), vz = (int)(tq / Ho);
  const T* xn = x + (long)n * Di * Hi * Wi;
  float in[27];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int z = vz * stride + dz - 1, yy = vy * stride + dy - 1, xx = vx * stride + dx - 1;
        const bool ok = z >= 0 && z < Di && yy >= 0 && yy < Hi && xx >= 0 && xx < Wi;
        const float t = to_f32<T>(xn[((long)min(max(z, 0), Di - 1) * Hi + min(max(yy, 0), Hi - 1)) * Wi + min(max(xx, 0), Wi - 1)]);
        in[(dz * 3 + dy) * 3 + dx] = ok ? t : 0.f;
      }
  T* yo = y + v * C_out + oc0;
#pragma unroll
  for (int ob = 0; ob < OCT; ob += 8) {
    if (oc0 + ob >= C_out) break;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (bias && oc0 + ob + q < C_out) ? bias[oc0 + ob + q] : 0.f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if constexpr (OCT >= 8) {
        const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(&wl[tap * OCT + ob]);
        const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(&wl[tap * OCT + ob + 4]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q] = fmaf(in[tap], w0[q], acc[q]); acc[4 + q] = fmaf(in[tap], w1[q], acc[4 + q]); }
      } else {
#pragma unroll
        for (int q = 0; q < OCT; ++q) acc[q] = fmaf(in[tap], wl[tap * OCT + q], acc[q]);
      }
    }
    if (OCT >= 8 && (C_out % 8) == 0) {
      VecIO<T, 8>::store(yo + ob, acc);
    } else {
#pragma unroll
      for (int q = 0; q < (OCT < 8 ? OCT : 8); ++q)
        if (oc0 + ob + q < C_out) yo[ob + q] = from_f32<T>(acc[q]);
    }
  }
}

// -> true when the launch was made: one input channel, 3^3 taps, stride 1 ('same') or 2 (pad 1), no fused pre-activation / residual
bool conv_c1_stencil_try(const void* x, const void* wp, const float* bias, const float* ab, int act_in, const EpiParams& e, int N,
                         int Do, int Ho, int Wo, int Di, int Hi, int Wi, int C_in, int C_out, int kd, int kh, int kw, int stride, int pad,
                         int dtype, hipStream_t s) {
  if (C_in != 1 || kd != 3 || kh != 3 || kw != 3 || pad != 1 || (stride != 1 && stride != 2) || ab || act_in != PYTC_ACT_NONE ||
      e.res_mode != PYTC_RES_NONE || tuning_get("conv_c1_stencil", 1) == 0)
    return false;
  if (stride == 1 && (Do != Di || Ho != Hi || Wo != Wi)) return false;
  if (stride == 2 && (Do != (Di + 1) / 2 || Ho != (Hi + 1) / 2 || Wo != (Wi + 1) / 2)) return false;
  const long total = (long)N * Do * Ho * Wo;
  const int oct = C_out >= 32 ? 32 : (C_out >= 8 ? 8 : (C_out > 1 ? 4 : 1));
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)((C_out + oct - 1) / oct));
#define PYTC_C1(TT, OCTV) hipLaunchKernelGGL((conv3d_c1_stencil_kernel<TT, OCTV>), grid, dim3(256), 0, s, (const TT*)x, (const TT*)wp, bias, \
                                             (TT*)e.y, N, Do, Ho, Wo, Di, Hi, Wi, C_out, stride)
  if (dtype == PYTC_BF16) { if (oct == 32) PYTC_C1(bf16_t, 32); else if (oct == 8) PYTC_C1(bf16_t, 8); else if (oct == 4) PYTC_C1(bf16_t, 4); else PYTC_C1(bf16_t, 1); }
  else { if (oct == 32) PYTC_C1(float, 32); else if (oct == 8) PYTC_C1(float, 8); else if (oct == 4) PYTC_C1(float, 4); else PYTC_C1(float, 1); }
#undef PYTC_C1
  return true;
}

// ---- LDS-tiled form (bf16, C_in % 8 == 0) ------------------------------------------------------------------------------
// A workgroup owns a 4 x 8 x 16 (z, y, x) block of output voxels and MT*16 output channels.  Per chunk of KC input
// channels it stages the haloed input block ONCE into LDS with the pre-activation f already applied (the direct kernel
// above re-loads and re-activates every voxel once per tap), zero outside the volume, then runs the GEMM with the
// reduction index flattened over (tap, channel-in-chunk): group g covers 32 consecutive flattened indices, lane group
// kb its 8-channel slice of ONE tap, so a B fragment is one ds_read_b128 at  voxel(lane) + koff[g][kb]  and narrow
// layers (C_in = 8 / 16) fill the K = 32 of an MFMA with 4 / 2 taps instead of zero padding.  wave = z plane of the
// block, N tile = one 16-voxel x row.  Weights: packed [mtile][group][lane][8] (conv3d_pack_flat_kernel), streamed
// from L1/L2 one group ahead of the MFMAs.
constexpr int CT_TZ = 4, CT_TY = 8, CT_TX = 16;
struct ConvTile { int KC, nchunks, G, tzh, tyh, txh, tiles_z, tiles_y, tiles_x; int lds_total; };     // lds_total: dynamic LDS bytes of the launch
// the eight phases of a stride-2 transposed gather in ONE launch (blockIdx.z = phase = 4a + 2b + c): per phase the tap extents are
// (1 + a, 1 + b, 1 + c), the groups per chunk G and the element offset of its weight image differ; n = 0: an ordinary conv
struct ConvPhases { int n; int probe; };      // probe (measurements only, wrong results): 1 = no matrix loop, 2 = no staging loads, 3 = no epilogue

// channels per staged chunk: the whole (narrow) layer when it fits one chunk, else the widest divisor among 32 / 16 / 8
static __host__ __device__ inline int conv_kc(int C_in) {
  return C_in <= 32 ? C_in : (C_in % 32 == 0 ? 32 : (C_in % 16 == 0 ? 16 : 8));
}

template <int MT>
__global__ void __launch_bounds__(256, 2)
conv3d_tile_kernel(ConvParams p, ConvTile t, ConvPhases ps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int NT = CT_TY;
  // per-launch or per-phase geometry as wave-uniform scalars (a kernel argument indexed at run time would be copied to scratch)
  int kd = p.kd, kh = p.kh, kw = p.kw, oz = p.oz, oy = p.oy, ox = p.ox, tG = t.G, tzh = t.tzh, tyh = t.tyh, txh = t.txh;
  long woff = 0;
  if (ps.n) {
    const int phz = blockIdx.z, a = (phz >> 2) & 1, b = (phz >> 1) & 1, c = phz & 1;
    kd = 1 + a; kh = 1 + b; kw = 1 + c;
    oz = a; oy = b; ox = c;
    tzh = CT_TZ + a; tyh = CT_TY + b; txh = CT_TX + c;
    // groups per chunk and image offset of the phase, from the rule of conv_tile_plan / pytc_convT3d_phase_plan (scalar arithmetic)
    tG = (kd * kh * kw * t.KC + 31) / 32;
    for (int k = 0; k < phz; ++k) {
      const int gk = ((1 + ((k >> 2) & 1)) * (1 + ((k >> 1) & 1)) * (1 + (k & 1)) * t.KC + 31) / 32;
      woff += (long)p.MTt * t.nchunks * gk * 64 * 8;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, kb = lane >> 4;
  const int P = t.KC * 2;                                   // LDS bytes per staged voxel
  const int tile_vox = tzh * tyh * txh;
  const int ZOFF = tile_vox * P;                            // 16 zero bytes: the operand of padded K slots
  int* koff = reinterpret_cast<int*>(lds + ZOFF + 16);      // [G][4]
  // block -> (n, z tile, y tile, x tile); blockIdx.y = group of MT output-channel tiles
  long b = blockIdx.x;
  const int bx = (int)(b % t.tiles_x); b /= t.tiles_x;
  const int by = (int)(b % t.tiles_y); b /= t.tiles_y;
  const int bz = (int)(b % t.tiles_z);
  const int n = (int)(b / t.tiles_z);
  const int z0 = bz * CT_TZ, y0 = by * CT_TY, x0 = bx * CT_TX;
  const int mt0 = blockIdx.y * MT;
  const int pd = p.pd, ph = p.ph, pw = p.pw;
  const int ntap = kd * kh * kw;
  const long rps = (long)p.D * p.H * p.W;
  const bf16_t* xn = reinterpret_cast<const bf16_t*>(p.x) + (long)n * rps * p.C_in;

  if (threadIdx.x < 4) reinterpret_cast<int*>(lds + ZOFF)[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < tG * 4; i += 256) {
    const int idx = i * 8;
    const int tap = idx / t.KC, c = idx % t.KC;
    int off = -1;
    if (tap < ntap) {
      const int dx = tap % kw, tt = tap / kw;
      const int dy = tt % kh, dz = tt / kh;
      off = ((dz * tyh + dy) * txh + dx) * P + c * 2;
    }
    koff[i] = off;
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const bf16_t*>(p.wp) + woff);
  const int Gtot = t.nchunks * tG;
  const int CH = t.KC / 8;                                  // 16-byte pieces per staged voxel
  const int base0 = (wave * tyh * txh + r) * P;         // voxel (z = wave, y = 0, x = r) of the block
  const int nt_step = txh * P;

  // Round 6: with MT = 1 (the deep, small layers: few workgroups, each a serial chain of chunks) the NEXT chunk's pieces are requested
  // into registers before the matrix loop of the current one and written to LDS after it -- a chunk costs max(staging latency, matrix
  // loop) instead of their sum (256 -> 256 at 2 x 3 x 32 x 32: 8 chunks of 17 loads + 216 matrix instructions each, 164 us).  The piece
  // descriptors (global offset of chunk 0, validity) are formed ONCE per thread: no address arithmetic per chunk.  MT = 4 keeps the
  // plain order: its 128 accumulator registers leave no room for 20 staged pieces, and those launches fill the chip.
  constexpr bool PREF = false;      // measured (round 6): 167.6 against 164.5 us for 256 -> 256 -- the chunk's staging was not what the workgroup waited for
  constexpr int MAXP = 20;                                  // 16-byte pieces per thread: the tile is <= 80 KB
  typedef unsigned int cu4_t __attribute__((ext_vector_type(4)));      // (an array of HIP's uint4 struct goes to scratch: DESIGN.md 4.8)
  cu4_t pf[PREF ? MAXP : 1];
  int goff[PREF ? MAXP : 1];                                // element offset of the piece in chunk 0 (from xn); -1: outside the volume / the tile
  const int npiece = tile_vox * CH;
  if constexpr (PREF) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int i = threadIdx.x + 256 * k;
      goff[k] = -1;
      if (i < npiece) {
        const int vox = i / CH, piece = i % CH;
        const int tx = vox % txh, tq = vox / txh;
        const int ty = tq % tyh, tz = tq / tyh;
        const int z = z0 + tz - pd, y = y0 + ty - ph, x = x0 + tx - pw;
        if (z >= 0 && z < p.D && y >= 0 && y < p.H && x >= 0 && x < p.W) goff[k] = (int)((((long)z * p.H + y) * p.W + x) * p.C_in) + piece * 8;
      }
    }
  }
  auto fetch = [&](int ck) {                                // pieces of chunk ck -> registers (zero outside the volume)
    if constexpr (PREF) {
      const bf16_t* xc = xn + ck * t.KC;
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        cu4_t v = {0u, 0u, 0u, 0u};
        if (goff[k] >= 0) v = *reinterpret_cast<const cu4_t*>(xc + goff[k]);
        pf[k] = v;
      }
    }
  };
  auto preact = [&](cu4_t v, int c) -> cu4_t {              // the fused pre-activation of 8 channels starting at c
    f32x8_t f = __builtin_convertvector(__builtin_bit_cast(bf16x8_t, v), f32x8_t);
    if (p.ab != nullptr) {
      const float* av = p.ab + ((long)n * 2 + 0) * p.C_in + c;
      const float* bv = p.ab + ((long)n * 2 + 1) * p.C_in + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaf(f[j], av[j], bv[j]);
    }
    if (p.act_in != PYTC_ACT_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = pre_act(f[j], p.act_in, p.act_param);
    }
    return __builtin_bit_cast(cu4_t, __builtin_convertvector(f, bf16x8_t));
  };
  const bool has_pre = p.ab != nullptr || p.act_in != PYTC_ACT_NONE;
  fetch(0);

  for (int ck = 0; ck < t.nchunks; ++ck) {
    if (ck > 0) __syncthreads();                            // everyone is done reading the previous chunk
    const int c0 = ck * t.KC;
    if constexpr (PREF) {
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < npiece) {
          cu4_t v = pf[k];
          // (zero padding applies to f(X): a voxel outside the volume stays zero)
          if (has_pre && goff[k] >= 0) v = preact(v, c0 + (i % CH) * 8);
          *reinterpret_cast<cu4_t*>(lds + i * 16) = v;      // = lds + vox * P + piece * 16
        }
      }
      __syncthreads();
      if (ck + 1 < t.nchunks) fetch(ck + 1);
    } else {
      for (int i = threadIdx.x; i < tile_vox * CH; i += 256) {
        const int vox = i / CH, piece = i % CH;
        const int tx = vox % txh, tq = vox / txh;
        const int ty = tq % tyh, tz = tq / tyh;
        const int z = z0 + tz - pd, y = y0 + ty - ph, x = x0 + tx - pw;
        cu4_t v = {0u, 0u, 0u, 0u};
        if (ps.probe != 2 && z >= 0 && z < p.D && y >= 0 && y < p.H && x >= 0 && x < p.W) {
          const int c = c0 + piece * 8;
          v = *reinterpret_cast<const cu4_t*>(xn + (((long)z * p.H + y) * p.W + x) * p.C_in + c);
          if (has_pre) v = preact(v, c);
        }
        *reinterpret_cast<cu4_t*>(lds + vox * P + piece * 16) = v;
      }
      __syncthreads();
    }

    // Weight fragments stream from L1 / L2 (one 16-byte load per lane, group and output tile).  With ONE group in flight (rounds 1-5) a
    // workgroup of the deep, small layers -- MT = 1: eight matrix instructions per fragment, ~50 ns -- waited a whole L2 round trip
    // per group (256 -> 256 at 2 x 3 x 32 x 32: 1 728 groups per wave, 164 us for 12 us of matrix instructions).  A ring of AD groups ahead
    // (6 / 3 / 1 for MT = 1 / 2 / 4: what the accumulators leave room for); ring slots are compile-time (the group loop advances AD at a time).
    constexpr int AD = MT == 1 ? 6 : (MT == 2 ? 4 : 1);
    // ... and (MT <= 2) the B fragments of the NEXT group are read from LDS while the matrix instructions of the current one run: the
    // small launches put ONE wave on a SIMD, so the koff -> address -> ds_read_b128 chain of a group (two LDS round trips) was paid in
    // full 216 times per wave (matrix loop of 256 -> 256: 80 us for 12 us of matrix instructions, tools/r06_conv_tile_probe.py).  Buffer
    // parity is the ring slot's (AD is even).
    constexpr bool BDB = MT <= 2;
    bf16x8_t ring[AD][MT];
    auto wfrag = [&](int mt, int g) -> bf16x8_t {
      return (mt0 + mt < p.MTt && g < tG) ? wp[((long)(mt0 + mt) * Gtot + ck * tG + g) * 64 + lane] : bf16x8_t{};
    };
    auto bfrag = [&](int ko, int nt) -> bf16x8_t {
      return *reinterpret_cast<const bf16x8_t*>(lds + (ko < 0 ? ZOFF : base0 + nt * nt_step + ko));
    };
#pragma unroll
    for (int d = 0; d < AD; ++d)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ring[d][mt] = wfrag(mt, d);
    bf16x8_t bfb[BDB ? 2 : 1][NT];
    int ko1 = 0;                                            // koff of group g + 1
    if constexpr (BDB) {
      const int ko0 = koff[kb];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bfb[0][nt] = bfrag(ko0, nt);
      ko1 = tG > 1 ? koff[4 + kb] : -1;
    }
    for (int g0 = 0; g0 < (ps.probe == 1 ? 0 : tG); g0 += AD) {
#pragma unroll
      for (int d = 0; d < AD; ++d) {
        const int g = g0 + d;
        if (g < tG) {
          bf16x8_t af[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) { af[mt] = ring[d][mt]; ring[d][mt] = wfrag(mt, g + AD); }
          if constexpr (BDB) {
            const int cur = d & 1, nxt = cur ^ 1;      // (d is an unrolled loop index: compile-time after unrolling)
            const int ko2 = g + 2 < tG ? koff[(g + 2) * 4 + kb] : -1;
            if (g + 1 < tG) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) bfb[nxt][nt] = bfrag(ko1, nt);
            }
            ko1 = ko2;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfb[cur][nt], acc[mt][nt], 0, 0, 0);
          } else {
            const int ko = koff[g * 4 + kb];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const bf16x8_t bf = bfrag(ko, nt);
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf, acc[mt][nt], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  const int z = z0 + wave;
  const int x = x0 + r;
  // Round 6 (second session): the results turn round in LDS before they are stored.  A lane ends
  // with 4 channels of one voxel -- 8-byte stores 2 * C_out bytes apart: at 24 -> 24 k133 on 2 x 18 x 256 x 256 (RSUNet's stock
  // full-resolution layer) the epilogue was 93 of the launch's 183 us (probe 3 of tools/r06_conv_tile_probe.py) for 113 MB of output.
  // Each wave writes its z plane's (bias + activation)-finished fp32 values into a wave-private [y][x][channel] image (row pitch
  // CW + 4 floats: the 16 x positions of a float4 column fall on 16 distinct bank groups) over the input tile, which every wave is done
  // with, and reads whole voxel rows back: a lane takes 8 consecutive channels, adds the residual's 16 bytes, rounds ONCE and stores 16
  // bytes -- an x row of the tile is one contiguous 32 * C_out-byte run in HBM.  Same values, same single rounding as finish_and_store.
  {
    // (MT = 4: two passes of two channel tiles each through the same image -- 64-byte runs per voxel instead of four 8-byte pieces)
    constexpr int TP = MT >= 2 ? 2 : 1;                       // channel tiles per pass
    constexpr int CW = TP * 16;
    if (!p.om && (p.e.res_mode == PYTC_RES_NONE || p.e.res_mode == PYTC_RES_ADD) && (p.C_out % 8) == 0 && ps.probe != 3 &&
        4 * NT * 16 * (CW + 4) * 4 <= t.lds_total) {
      __syncthreads();                                        // the input tile is dead: every wave has left the matrix loop
      float* img = reinterpret_cast<float*>(lds) + wave * (NT * 16 * (CW + 4));
      const int zq = z0 + wave;
      bf16_t* yn = reinterpret_cast<bf16_t*>(p.e.y) + (long)n * rps * p.C_out;
      const bf16_t* resn = p.e.res_mode == PYTC_RES_ADD ? reinterpret_cast<const bf16_t*>(p.e.res) + (long)n * rps * p.C_out : nullptr;
#pragma unroll
      for (int ps_ = 0; ps_ < MT / TP; ++ps_) {
        const int c_base = (mt0 + ps_ * TP) * 16;
        if (c_base >= p.C_out) break;                         // (workgroup-uniform)
        const int cw_live = p.C_out - c_base < CW ? p.C_out - c_base : CW;      // multiple of 8 (C_out is)
        if (ps_ > 0) { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#pragma unroll
        for (int tt = 0; tt < TP; ++tt) {
          const int mt = ps_ * TP + tt;
          const int o0 = (mt0 + mt) * 16 + kb * 4;
          float bo[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) bo[rr] = (p.bias && o0 + rr < p.C_out) ? p.bias[o0 + rr] : 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4_t v;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = apply_act(acc[mt][nt][rr] + bo[rr], p.act_out);
            *reinterpret_cast<f32x4_t*>(img + (nt * 16 + r) * (CW + 4) + tt * 16 + kb * 4) = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (zq >= p.D) continue;
        const int cpv = cw_live / 8;                          // 16-byte output chunks per voxel
        const int nchunk = 16 * cpv;                          // ... per x row of the tile
        for (int q = lane; q < nchunk; q += 64) {
          const int xq = q / cpv, cq = (q - xq * cpv) * 8;
          if (x0 + xq >= p.W) continue;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (y0 + nt >= p.H) break;
            const float* src = img + (nt * 16 + xq) * (CW + 4) + cq;
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(src), hi = *reinterpret_cast<const f32x4_t*>(src + 4);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const long off = ((((long)zq * p.H + (y0 + nt)) * p.W + (x0 + xq)) * p.C_out) + c_base + cq;
            if (resn) {
              float rv[8];
              VecIO<bf16_t, 8>::load(resn + off, rv);
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] += rv[i];
            }
            VecIO<bf16_t, 8>::store(yn + off, v);
          }
        }
      }
      return;
    }
  }
  if (z >= p.D || x >= p.W || ps.probe == 3) return;      // (probe 3: no epilogue)
  // output row of (z, y0 + nt, x): one 64-bit base and a 32-bit step (the 2x output map of a phase launch steps two output rows)
  const long row0 = p.om ? ((long)(2 * z + oz) * p.Ho + (2 * y0 + oy)) * p.Wo + (2 * x + ox) : ((long)z * p.H + y0) * p.W + x;
  const int row_step = p.om ? 2 * p.Wo : p.W;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int o0 = (mt0 + mt) * 16 + kb * 4;
    if (o0 >= p.C_out) continue;
    float bo[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) bo[rr] = (p.bias && o0 + rr < p.C_out) ? p.bias[o0 + rr] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int y = y0 + nt;
      if (y >= p.H) continue;
      float v[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) v[rr] = apply_act(acc[mt][nt][rr] + bo[rr], p.act_out);
      finish_and_store<bf16_t, 4>(v, p.e, n, row0 + (long)nt * row_step, o0);
    }
  }
}

// packed [mtile][chunk][group][lane][8]: lane (r, kb), element j <-> flattened index q = group*32 + kb*8 + j of the chunk's
// (tap, channel-in-chunk) reduction: tap = q / KC, channel = chunk*KC + q % KC; zero where tap >= ntap
__global__ void __launch_bounds__(256)
conv3d_pack_flat_kernel(const float* __restrict__ w, int C_out, int C_in, int ntap, bf16_t* __restrict__ packed, int KC,
                        int nchunks, int G, long total, long s_o, long s_c, int flip) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % 8);
  long q = i / 8;
  const int lane = (int)(q % 64); q /= 64;
  const int g = (int)(q % G); q /= G;
  const int ck = (int)(q % nchunks);
  const int mt = (int)(q / nchunks);
  const int o = mt * 16 + (lane & 15);
  const int f = g * 32 + (lane >> 4) * 8 + j;
  const int tap = f / KC, c = ck * KC + f % KC;
  float v = 0.f;
  if (o < C_out && tap < ntap) v = w[o * s_o + c * s_c + (flip ? ntap - 1 - tap : tap)];
  packed[i] = from_f32<bf16_t>(v);
}

static bool conv_tile_plan(int dtype, int C_in, int kd, int kh, int kw, ConvTile& t, size_t& lds_bytes) {
  if (dtype != PYTC_BF16 || C_in % 8 != 0) return false;
  t.KC = conv_kc(C_in);
  t.lds_total = 0;
  t.nchunks = C_in / t.KC;
  t.G = (kd * kh * kw * t.KC + 31) / 32;
  t.tzh = CT_TZ + kd - 1; t.tyh = CT_TY + kh - 1; t.txh = CT_TX + kw - 1;
  lds_bytes = (size_t)t.tzh * t.tyh * t.txh * t.KC * 2 + 16 + (size_t)t.G * 16;
  return lds_bytes <= 160 * 1024 / 2;                      // two workgroups per CU
}

template <typename TW>
__global__ void __launch_bounds__(256)
conv3d_pack_kernel(const float* __restrict__ w, int C_out, int C_in, int ntap, TW* __restrict__ packed, int KG,
                   long total, long s_o, long s_c, int flip) {
  // source: PyTorch layout [C_out][C_in][kd][kh][kw]  (tap fastest)
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

// Every conv-weight image of a model in ONE launch (the per-step repack of a training run: 64 launches of ~7 us each per RSUNet
// step otherwise).  table: n_items rows of 16 int64 = { w, packed, s_o, s_c, first block, elements, C_out, C_in, ntap,
// kind (0: [mtile][tap][kgroup][lane][EPL] of conv3d_pack_kernel / conv3d_pack_direct_kernel, 1: the flat chunked layout of
// conv3d_pack_flat_kernel), fp32 image (else bf16), flip, KG | KC, nchunks, G, 0 }; a 256-thread block belongs to one row.
__global__ void __launch_bounds__(256)
conv3d_pack_multi_kernel(const long* __restrict__ table, int n_items) {
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {                                  // last row whose first block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * 16 + 4] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long* r = table + (long)lo * 16;
  const long i = ((long)blockIdx.x - r[4]) * 256 + threadIdx.x;
  if (i >= r[5]) return;
  const float* w = reinterpret_cast<const float*>(r[0]);
  const long s_o = r[2], s_c = r[3];
  const int C_out = (int)r[6], C_in = (int)r[7], ntap = (int)r[8], kind = (int)r[9], f32 = (int)r[10], flip = (int)r[11];
  int o, k, tap;
  if (kind == 2) {
    // one phase (bits a b c = z y x parity of the output voxel) of a k 3 / stride 2 / pad 1 transposed gather as a stride-1 conv with
    // (1 + a) x (1 + b) x (1 + c) taps on the input grid: out[2i] = w[1] x[i];  out[2i + 1] = w[2] x[i] + w[0] x[i + 1] per axis --
    // flat chunked layout of kind 1 with the phase's tap count, the source tap read off the sub-tap (d = 0 -> x[i], d = 1 -> x[i + 1])
    const int KC = (int)r[12], nchunks = (int)r[13], G = (int)r[14], phase = flip;
    const int j = (int)(i % 8);
    long q = i / 8;
    const int lane = (int)(q % 64); q /= 64;
    const int g = (int)(q % G); q /= G;
    const int ck = (int)(q % nchunks);
    const int mt = (int)(q / nchunks);
    o = mt * 16 + (lane & 15);
    const int f = g * 32 + (lane >> 4) * 8 + j;
    const int st = f / KC;                       // sub-tap of the phase, (dz, dy, dx) with extents (1 + a, 1 + b, 1 + c)
    k = ck * KC + f % KC;
    const int a = (phase >> 2) & 1, b = (phase >> 1) & 1, c = phase & 1;
    float v = 0.f;
    if (o < C_out && k < C_in && st < ntap) {
      const int dx = st % (1 + c), t2 = st / (1 + c);
      const int dy = t2 % (1 + b), dz = t2 / (1 + b);
      const int tz = a ? (dz ? 0 : 2) : 1, ty = b ? (dy ? 0 : 2) : 1, tx = c ? (dx ? 0 : 2) : 1;
      v = w[o * s_o + k * s_c + (tz * 3 + ty) * 3 + tx];
    }
    reinterpret_cast<bf16_t*>(r[1])[i] = from_f32<bf16_t>(v);
    return;
  }
  if (kind == 1) {
    const int KC = (int)r[12], nchunks = (int)r[13], G = (int)r[14];
    const int j = (int)(i % 8);
    long q = i / 8;
    const int lane = (int)(q % 64); q /= 64;
    const int g = (int)(q % G); q /= G;
    const int ck = (int)(q % nchunks);
    const int mt = (int)(q / nchunks);
    o = mt * 16 + (lane & 15);
    const int f = g * 32 + (lane >> 4) * 8 + j;
    tap = f / KC;
    k = ck * KC + f % KC;
  } else {
    const int KG = (int)r[12], EPL = f32 ? 4 : 8, KSTEP = f32 ? 16 : 32;
    const int j = (int)(i % EPL);
    long t = i / EPL;
    const int lane = (int)(t % 64); t /= 64;
    const int kg = (int)(t % KG); t /= KG;
    tap = (int)(t % ntap);
    const int mt = (int)(t / ntap);
    o = mt * 16 + (lane & 15);
    k = kg * KSTEP + (lane >> 4) * EPL + j;
  }
  float v = 0.f;
  if (o < C_out && k < C_in && tap < ntap) v = w[o * s_o + k * s_c + (flip ? ntap - 1 - tap : tap)];
  if (f32) reinterpret_cast<float*>(r[1])[i] = v;
  else reinterpret_cast<bf16_t*>(r[1])[i] = from_f32<bf16_t>(v);
}

template <int MT>
static void launch_conv_tile_mt(const ConvParams& p, const ConvTile& t, size_t lds_bytes, dim3 grid, hipStream_t s) {
  // dynamic LDS above 64 KB needs the opt-in, once per kernel and device
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3d_tile_kernel<MT>), 80 * 1024, "conv3d_tile")) return;
  ConvTile tt = t;
  // narrow layers turn their results round in LDS (the kernel's epilogue): room for the waves' fp32 [8][16][MT * 16 + 4] images
  size_t total = lds_bytes;
  const int knob = tuning_get("conv_tile_lds_epilogue", 3);      // bit 0: MT <= 2, bit 1: MT = 4 (two passes)
  const bool on = MT <= 2 ? (knob & 1) != 0 : (knob & 2) != 0;
  if (on) {
    const size_t need = (size_t)4 * CT_TY * 16 * ((MT >= 2 ? 32 : 16) + 4) * 4;
    if (need <= 80 * 1024 && need > total) total = need;
  }
  tt.lds_total = on ? (int)total : 0;
  hipLaunchKernelGGL((conv3d_tile_kernel<MT>), grid, dim3(256), total, s, p, tt, ConvPhases{0, tuning_get("conv_tile_probe", 0)});
}

template <int MT>
static void launch_conv_phases_mt(const ConvParams& p, const ConvTile& t, const ConvPhases& ps, size_t lds_bytes, dim3 grid, hipStream_t s) {
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&conv3d_tile_kernel<MT>), 80 * 1024, "conv3d_tile")) return;
  hipLaunchKernelGGL((conv3d_tile_kernel<MT>), grid, dim3(256), lds_bytes, s, p, t, ps);
}

static void launch_conv_tile(const ConvParams& p, ConvTile t, size_t lds_bytes, hipStream_t s) {
  t.tiles_z = (p.D + CT_TZ - 1) / CT_TZ; t.tiles_y = (p.H + CT_TY - 1) / CT_TY; t.tiles_x = (p.W + CT_TX - 1) / CT_TX;
  int MT = p.MTt >= 4 ? 4 : (p.MTt >= 2 ? 2 : 1);
  // deep levels (2 x 18 x 20 x 20 voxels = 60 spatial tiles): 64 output channels per workgroup leave 120 workgroups for 256 CUs;
  // fewer output tiles per workgroup (every workgroup stages the same input tile: L2 hits) until one per CU exists (more costs level 1 its 64-channel reuse: 40 -> 55 us)
  const long spatial = (long)p.N * t.tiles_z * t.tiles_y * t.tiles_x;
  if (tuning_get("conv_tile_small_mt", 1))
    while (MT > 1 && spatial * ((p.MTt + MT - 1) / MT) < 256) MT >>= 1;
  dim3 grid((unsigned)spatial, (unsigned)((p.MTt + MT - 1) / MT));
  switch (MT) {
    case 1: launch_conv_tile_mt<1>(p, t, lds_bytes, grid, s); break;
    case 2: launch_conv_tile_mt<2>(p, t, lds_bytes, grid, s); break;
    default: launch_conv_tile_mt<4>(p, t, lds_bytes, grid, s); break;
  }
}

template <typename TI, typename TW, typename TO>
static void launch_conv(const ConvParams& p, hipStream_t s) {
  constexpr int NT = 4;
  const long rps = (long)p.D * p.H * p.W;
  const size_t thin_lds = (size_t)p.kd * p.kh * p.kw * p.C_in * 16 * sizeof(float);
  if (p.C_in <= 4 && thin_lds <= 48 * 1024 && tuning_get("conv_thin_in", 1) != 0) {
    dim3 grid((unsigned)((rps + 255) / 256), (unsigned)p.MTt, (unsigned)p.N);
    hipLaunchKernelGGL((conv3d_thin_in_kernel<TI, TW, TO>), grid, dim3(256), thin_lds, s, p);
    return;
  }
  const int MT = p.MTt >= 4 ? 4 : (p.MTt >= 2 ? 2 : 1);
  dim3 grid((unsigned)((rps + 4L * NT * 16 - 1) / (4L * NT * 16)), (unsigned)((p.MTt + MT - 1) / MT), (unsigned)p.N);
  dim3 block(256);
  switch (MT) {
    case 1: hipLaunchKernelGGL((conv3d_kernel<TI, TW, TO, 1, NT>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((conv3d_kernel<TI, TW, TO, 2, NT>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((conv3d_kernel<TI, TW, TO, 4, NT>), grid, block, 0, s, p); break;
  }
}

}  // namespace pytc

using namespace pytc;

static int kstep_of(int dtype) { return dtype == PYTC_BF16 ? 32 : 16; }

extern "C" int64_t pytc_conv3d_packed_elems(int C_out, int C_in, int kd, int kh, int kw, int dtype) {
  if (C_out < 1 || C_in < 1 || kd < 1 || kh < 1 || kw < 1 || (dtype != PYTC_F32 && dtype != PYTC_BF16)) return -1;
  int ks = kstep_of(dtype);
  const int64_t direct = (int64_t)((C_out + 15) / 16) * 16 * kd * kh * kw * ((C_in + ks - 1) / ks) * ks;
  ConvTile t; size_t lds_bytes;
  if (!conv_tile_plan(dtype, C_in, kd, kh, kw, t, lds_bytes)) return direct;
  const int64_t flat = (int64_t)((C_out + 15) / 16) * t.nchunks * t.G * 64 * 8;     // every chunk pads its K to 32
  return flat > direct ? flat : direct;
}

// C_out / C_in describe the conv the packed image is FOR; (s_o, s_c, flip) say where element (o, c, tap) lives in `w`
static int pack_conv_weight(const float* w, int C_out, int C_in, int kd, int kh, int kw, void* packed, int dtype, long s_o,
                            long s_c, int flip, void* stream) {
  PYTC_REQUIRE(w && packed, "conv3d_pack_weight: null pointer");
  long total = pytc_conv3d_packed_elems(C_out, C_in, kd, kh, kw, dtype);
  PYTC_REQUIRE(total > 0, "conv3d_pack_weight: bad arguments");
  int KG = (C_in + kstep_of(dtype) - 1) / kstep_of(dtype);
  dim3 grid(ceil_div(total, 256)), block(256);
  ConvTile t; size_t lds_bytes;
  if (conv_tile_plan(dtype, C_in, kd, kh, kw, t, lds_bytes)) {      // the LDS-tiled kernel's layout (same rule as the launch)
    const long tot2 = (long)((C_out + 15) / 16) * t.nchunks * t.G * 64 * 8;
    hipLaunchKernelGGL(conv3d_pack_flat_kernel, dim3(ceil_div(tot2, 256)), block, 0, (hipStream_t)stream, w, C_out, C_in,
                       kd * kh * kw, (bf16_t*)packed, t.KC, t.nchunks, t.G, tot2, s_o, s_c, flip);
  } else if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(conv3d_pack_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, kd * kh * kw,
                       (bf16_t*)packed, KG, total, s_o, s_c, flip);
  else
    hipLaunchKernelGGL(conv3d_pack_kernel<float>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, kd * kh * kw,
                       (float*)packed, KG, total, s_o, s_c, flip);
  PYTC_LAUNCH_CHECK("conv3d_pack_weight");
  return PYTC_OK;
}

extern "C" int pytc_conv3d_pack_weight(const float* w, int C_out, int C_in, int kd, int kh, int kw, void* packed,
                                       int dtype, void* stream) {
  const long ntap = (long)kd * kh * kw;
  return pack_conv_weight(w, C_out, C_in, kd, kh, kw, packed, dtype, (long)C_in * ntap, ntap, 0, stream);
}

// weights of the data-gradient conv (C_out -> C_in channels, taps mirrored) straight from the forward weight
// w [C_out][C_in][kd][kh][kw]: element (o' = c, c' = o, tap') = w[o][c][ntap - 1 - tap']
extern "C" int pytc_conv3d_pack_weight_dgrad(const float* w, int C_out, int C_in, int kd, int kh, int kw, void* packed,
                                             int dtype, void* stream) {
  const long ntap = (long)kd * kh * kw;
  return pack_conv_weight(w, C_in, C_out, kd, kh, kw, packed, dtype, ntap, (long)C_in * ntap, 1, stream);
}

/* the layout pytc_conv3d_pack_weight (direct = 0) / pytc_conv3d_pack_weight_direct (direct = 1) would write for a conv with these
   channel counts: out[0] = kind (0 tap-major, 1 flat chunked), out[1] = KG | KC, out[2] = nchunks, out[3] = G, out[4] = elements the
   pack writes (<= the *_packed_elems allocation).  For building the table of pytc_conv3d_pack_multi. */
extern "C" int pytc_conv3d_pack_plan(int C_out, int C_in, int kd, int kh, int kw, int dtype, int direct, int64_t* out) {
  PYTC_REQUIRE(out && C_out >= 1 && C_in >= 1 && kd >= 1 && kh >= 1 && kw >= 1 && (dtype == PYTC_F32 || dtype == PYTC_BF16),
               "conv3d_pack_plan: bad arguments");
  const int ks = kstep_of(dtype);
  ConvTile t; size_t lds_bytes;
  if (!direct && conv_tile_plan(dtype, C_in, kd, kh, kw, t, lds_bytes)) {
    out[0] = 1; out[1] = t.KC; out[2] = t.nchunks; out[3] = t.G;
    out[4] = (int64_t)((C_out + 15) / 16) * t.nchunks * t.G * 64 * 8;
  } else {
    out[0] = 0; out[1] = (C_in + ks - 1) / ks; out[2] = 0; out[3] = 0;
    out[4] = (int64_t)((C_out + 15) / 16) * 16 * kd * kh * kw * ((C_in + ks - 1) / ks) * ks;
  }
  return PYTC_OK;
}

/* table_dev: n_items x 16 int64 on the device (layout: conv3d_pack_multi_kernel), total_blocks = sum of ceil(elements / 256) */
extern "C" int pytc_conv3d_pack_multi(const int64_t* table_dev, int n_items, int64_t total_blocks, void* stream) {
  PYTC_REQUIRE(table_dev && n_items >= 1 && total_blocks >= 1 && total_blocks < (1LL << 31), "conv3d_pack_multi: bad arguments");
  hipLaunchKernelGGL(conv3d_pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long*>(table_dev), n_items);
  PYTC_LAUNCH_CHECK("conv3d_pack_multi");
  return PYTC_OK;
}

extern "C" int pytc_conv3d_fwd(const pytc_conv3d_args* a, void* stream) {
  PYTC_REQUIRE(a && a->x && a->w_packed && a->y, "conv3d: null pointer");
  PYTC_REQUIRE(a->N >= 1 && a->D >= 1 && a->H >= 1 && a->W >= 1 && a->C_in >= 1 && a->C_out >= 1, "conv3d: bad shape");
  PYTC_REQUIRE((a->kd & 1) && (a->kh & 1) && (a->kw & 1), "conv3d: kernel sizes must be odd ('same' padding)");
  PYTC_REQUIRE(a->dtype == PYTC_F32 || a->dtype == PYTC_BF16, "conv3d: bad dtype");
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || (a->res_mode == PYTC_RES_ADD && a->res), "conv3d: bad residual");
  ConvParams p;
  p.x = a->x; p.wp = a->w_packed; p.bias = a->bias; p.ab = a->ab;
  p.N = a->N; p.D = a->D; p.H = a->H; p.W = a->W; p.C_in = a->C_in; p.C_out = a->C_out;
  p.KG = (a->C_in + kstep_of(a->dtype) - 1) / kstep_of(a->dtype);
  p.MTt = (a->C_out + 15) / 16;
  p.kd = a->kd; p.kh = a->kh; p.kw = a->kw;
  p.act_in = a->act_in; p.act_param = a->act_param; p.act_out = PYTC_ACT_NONE;
  p.pd = a->kd / 2; p.ph = a->kh / 2; p.pw = a->kw / 2;
  p.om = p.oz = p.oy = p.ox = p.Ho = p.Wo = 0;
  p.e.res = a->res; p.e.res_low = nullptr; p.e.res_bias = nullptr; p.e.y = a->y;
  p.e.rps_out = (long)a->D * a->H * a->W; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode; p.e.nt = 0;
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  hipStream_t s = (hipStream_t)stream;
  ConvTile t; size_t lds_bytes;
  if (conv_c1_stencil_try(a->x, a->w_packed, a->bias, a->ab, a->act_in, p.e, a->N, a->D, a->H, a->W, a->D, a->H, a->W, a->C_in, a->C_out,
                          a->kd, a->kh, a->kw, 1, 1, a->dtype, s)) {}
  else if (conv_tile_plan(a->dtype, a->C_in, a->kd, a->kh, a->kw, t, lds_bytes)) launch_conv_tile(p, t, lds_bytes, s);
  else if (a->dtype == PYTC_F32) launch_conv<float, float, float>(p, s);
  else launch_conv<bf16_t, bf16_t, bf16_t>(p, s);
  PYTC_LAUNCH_CHECK("conv3d");
  return PYTC_OK;
}

// ---- stride-2 transposed gather (k 3, pad 1, output = 2 x input grid) as EIGHT stride-1 convs on the LDS-tiled kernel (round 6) ---------
// The gather form (conv3d_strided_kernels.hip) loads every operand fragment from global memory per tap with nothing in flight:
// convT3d_fwd[384 -> 64] ran at 11 TFLOP/s, and halving its tap iterations (phase-major rows) bought 4 %.  A phase
// (a, b, c) = parity of the output voxel is a stride-1 correlation with (1 + a)(1 + b)(1 + c) taps on the INPUT grid (forward halo of one
// voxel on the odd axes) that writes every second voxel per axis: conv3d_tile_kernel with explicit padding and the 2x output map, one
// launch per phase (27 taps over 8 phases: no zero work), the haloed input block staged once per chunk of input channels.
// Serves ConvTranspose3d(k 3, s 2, p 1, output_padding 1) and the data gradient of Conv3d(k 3, s 2, p 1) on even grids.
static bool convT_phase_plan(int dtype, int C_in, ConvTile (&t)[8], size_t (&lds)[8]) {
  for (int ph = 0; ph < 8; ++ph)
    if (!conv_tile_plan(dtype, C_in, 1 + ((ph >> 2) & 1), 1 + ((ph >> 1) & 1), 1 + (ph & 1), t[ph], lds[ph])) return false;
  return true;
}

/* out[0..7] = element offset of the phase images in one buffer, out[8] = total elements, out[9] = KC, out[10] = nchunks, out[11..18] = G */
extern "C" int pytc_convT3d_phase_plan(int C_out, int C_in, int dtype, int64_t* out) {
  PYTC_REQUIRE(out && C_out >= 1 && C_in >= 1, "convT3d_phase_plan: bad arguments");
  ConvTile t[8]; size_t lds[8];
  if (!convT_phase_plan(dtype, C_in, t, lds)) return PYTC_ERR_UNSUPPORTED;
  int64_t off = 0;
  for (int ph = 0; ph < 8; ++ph) {
    out[ph] = off;
    out[11 + ph] = t[ph].G;
    off += (int64_t)((C_out + 15) / 16) * t[ph].nchunks * t[ph].G * 64 * 8;
  }
  out[8] = off; out[9] = t[0].KC; out[10] = t[0].nchunks;
  return PYTC_OK;
}

extern "C" int pytc_convT3d_phase_supported(int C_out, int C_in, int dtype) {
  ConvTile t[8]; size_t lds[8];
  return (C_out >= 1 && C_in >= 1 && convT_phase_plan(dtype, C_in, t, lds) && tuning_get("convT_phase_tile", 1) != 0) ? 1 : 0;
}

/* a->D/H/W: the OUTPUT grid (= 2 x in_dims), a->w_packed: the eight phase images (pytc_convT3d_phase_plan offsets), a->kd = kh = kw = 3 */
extern "C" int pytc_convT3d_phase_fwd(const pytc_conv3d_args* a, const int32_t* in_dims, void* stream) {
  PYTC_REQUIRE(a && a->x && a->w_packed && a->y && in_dims, "convT3d_phase: null pointer");
  PYTC_REQUIRE(a->N >= 1 && a->C_in >= 1 && a->C_out >= 1 && a->dtype == PYTC_BF16, "convT3d_phase: bf16 only");
  PYTC_REQUIRE(a->kd == 3 && a->kh == 3 && a->kw == 3 && a->D == 2 * in_dims[0] && a->H == 2 * in_dims[1] && a->W == 2 * in_dims[2],
               "convT3d_phase: kernel 3, stride 2, padding 1, output grid = 2 x input grid");
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || (a->res_mode == PYTC_RES_ADD && a->res), "convT3d_phase: bad residual");
  ConvTile t[8]; size_t lds[8];
  PYTC_REQUIRE(convT_phase_plan(a->dtype, a->C_in, t, lds), "convT3d_phase: C_in % 8 != 0 or the tile does not fit LDS");
  hipStream_t s = (hipStream_t)stream;
  ConvParams p;
  p.x = a->x; p.wp = a->w_packed; p.bias = a->bias; p.ab = a->ab;
  p.N = a->N; p.D = in_dims[0]; p.H = in_dims[1]; p.W = in_dims[2]; p.C_in = a->C_in; p.C_out = a->C_out;
  p.KG = 0; p.MTt = (a->C_out + 15) / 16;
  p.kd = p.kh = p.kw = 1;                                  // (set per phase in the kernel)
  p.act_in = a->act_in; p.act_param = a->act_param; p.act_out = PYTC_ACT_NONE;
  p.pd = p.ph = p.pw = 0;
  p.om = 1; p.oz = p.oy = p.ox = 0; p.Ho = a->H; p.Wo = a->W;
  p.e.res = a->res; p.e.res_low = nullptr; p.e.res_bias = nullptr; p.e.y = a->y;
  p.e.rps_out = (long)a->D * a->H * a->W; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode; p.e.nt = 0;
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  ConvPhases ps;
  ps.n = 8; ps.probe = 0;
  size_t lds_max = 0;
  for (int ph = 0; ph < 8; ++ph)
    if (lds[ph] > lds_max) lds_max = lds[ph];
  ConvTile tt = t[7];
  tt.tiles_z = (p.D + CT_TZ - 1) / CT_TZ; tt.tiles_y = (p.H + CT_TY - 1) / CT_TY; tt.tiles_x = (p.W + CT_TX - 1) / CT_TX;
  int MT = p.MTt >= 4 ? 4 : (p.MTt >= 2 ? 2 : 1);
  const long spatial = (long)p.N * tt.tiles_z * tt.tiles_y * tt.tiles_x;
  while (MT > 1 && spatial * 8 * ((p.MTt + MT - 1) / MT) < 512) MT >>= 1;       // (as launch_conv_tile: enough workgroups for the chip)
  dim3 grid((unsigned)spatial, (unsigned)((p.MTt + MT - 1) / MT), 8);
  switch (MT) {
    case 1: launch_conv_phases_mt<1>(p, tt, ps, lds_max, grid, s); break;
    case 2: launch_conv_phases_mt<2>(p, tt, ps, lds_max, grid, s); break;
    default: launch_conv_phases_mt<4>(p, tt, ps, lds_max, grid, s); break;
  }
  PYTC_LAUNCH_CHECK("convT3d_phase");
  return PYTC_OK;
}

