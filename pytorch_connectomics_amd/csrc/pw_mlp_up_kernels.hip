// Fused MedNeXt UP block (bf16):  depthwise transposed 3x3x3 conv (stride 2)  ->  GroupNorm-apply -> 1x1 expand -> GELU ->
// 1x1 project  ->  + transposed-1x1 residual + encoder skip, with the depthwise output `t` NEVER in HBM.
//
// In the un-fused schedule `t` is the widest tensor of the network per byte of information: 2C channels at the HIGH
// resolution (1.44 GB per 8 windows at level 0), written by dwconvT3d_k3_cell and read back by pw_mlp -- while every one of
// its voxels is a function of at most 2x2x2 voxels of the 8x smaller low-resolution input.  Here the mixer's B operand is
// computed in the prologue straight from the low-resolution tensor (L1 / L2 resident, each voxel reused by 27 outputs):
//   * a wave's tile = 16 consecutive low-res cells along x  x  the 4 output positions (py, px) of one output z parity pz,
//     so (pz, py, px) -- hence the set of taps -- is UNIFORM across the wave for every voxel tile: no parity divergence, no
//     div / mod in the epilogue, and the residual gather (only the all-odd position reads the low-res transposed-1x1
//     result) is a uniform branch;
//   * a lane (cell r, channel chunk kb) loads the 2x2x2 (pz = 0) or 1x2x2 (pz = 1) low-res neighbours of its cell once per
//     32-channel k-step (16 B each) and forms the four positions' values with the SAME fp32 operation order as
//     dwconvT3d_k3_cell_kernel (bias first, taps in (a, b, d) order), rounds them to bf16 exactly as that kernel stores
//     them -- the fused path is bit-identical to the un-fused one given the same statistics;
//   * the 27 x C taps sit in LDS (the lanes of a channel chunk read the same address: broadcast);
//   * the rest is the fused mixer of pw_mlp_kernels.hip (paired-row packing, GEMM1 accumulator = GEMM2 operand, 16-byte
//     stores).
// GroupNorm statistics come from a statistics-only launch of dwconvT3d_k3_cell_kernel (y = nullptr: same kernel, same
// partial-sum tree, no 1.44 GB write).  HBM traffic per high-res voxel: skip (C_out) + y (C_out) + 1/8 of the low-res
// input instead of t write + t read + skip + y.
#include "pw_common.h"

namespace pytc {

struct MlpUpParams {
  const bf16_t* xlow;      // [N][D][H][W][C_in] low-resolution block input
  const float* taps;       // [27][C_in] depthwise transposed-conv weights, tap-major (kz, ky, kx)
  const float* b1;         // [C_in] depthwise bias (zeros when the conv has none)
  const float* ab;         // [N][2][C_in] GroupNorm affine of t
  const bf16x8_t* w2;
  const float* b2;
  const bf16x8_t* w3;
  const float* b3;
  const bf16_t* skip;      // [N][2D][2H][2W][C_out] encoder skip
  const bf16_t* res_low;   // [N][D][H][W][C_out] transposed-1x1 residual at low resolution (bias included) or NULL
  const float* res_bias;   // [C_out] value of that residual in the stride holes (zeros without a residual conv)
  bf16_t* y;               // [N][2D][2H][2W][C_out]
  int N, D, H, W, XS;      // low-res grid, XS = ceil(W / 16)
  int C_in, C_hid, C_out, HC;
  long tiles;
};

constexpr int up_waves_per_simd(int ks, int mo) {
  const int regs = ks * 4 * 4 + mo * 4 * 4 + 2 * 4 * 4 + 8 * 4 + (mo / 2) * 4 * 4;   // bact + acc2 + acc1 + xin + skip rows
  return regs <= 112 ? 3 : (regs <= 200 ? 2 : 1);
}

// The (position, neighbour, tap) sequence of one k-step in dwconvT3d_k3_cell_kernel's accumulation order: positions
// nt = (py, px) outermost, then neighbours a (z; only a = 1 when pz = 1), b >= py, d >= px.  Per axis an even output position
// reads (input 0, tap 2) and (input 1, tap 0), an odd one (input 1, tap 1).
struct UpStep { int nt, a, b, d, tap; bool first, last; };
constexpr int up_steps(int pz) { return pz ? 9 : 18; }
constexpr UpStep up_step(int pz, int s) {
  int i = 0;
  for (int nt = 0; nt < 4; ++nt) {
    const int py = nt >> 1, px = nt & 1;
    const int cnt = (2 - pz) * (2 - py) * (2 - px);
    int j = 0;
    for (int a = pz; a < 2; ++a)
      for (int b = py; b < 2; ++b)
        for (int d = px; d < 2; ++d, ++i, ++j)
          if (i == s) {
            const int kz = pz ? 1 : (a ? 0 : 2), ky = py ? 1 : (b ? 0 : 2), kx = px ? 1 : (d ? 0 : 2);
            return UpStep{nt, a, b, d, (kz * 3 + ky) * 3 + kx, j == 0, j == cnt - 1};
          }
  }
  return UpStep{0, 0, 0, 0, 0, false, false};
}

// One tile with the output z parity PZ known at compile time: the tap sets and the neighbour set are static, so the
// prologue is straight-line code (with a run-time parity hipcc keeps both variants' values live and spills ~70 registers).
template <int KS_IN, int MO, int GELU_MODE, int PZ>
__device__ __forceinline__ void up_tile(const MlpUpParams& p, const float* wl, int n, int mz, int my, int xs, int lane) {
  constexpr int NT = 4;
  constexpr int pz = PZ;
  const int r = lane & 15, kb = lane >> 4;
  const int mx = xs * 16 + r;
  const bool live = mx < p.W;
  const int mxc = live ? mx : p.W - 1;
  const int Ho = 2 * p.H, Wo = 2 * p.W;
  const int Pz = 2 * mz + pz;

  int orow[NT];                                               // output row inside the sample (< 2^31)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) orow[nt] = (Pz * Ho + 2 * my + (nt >> 1)) * Wo + 2 * mxc + (nt & 1);

  // ---- B operand of GEMM1: t = dwconvT(x_low) for the 4 positions of this lane's cell, normalised.
  // One 32-channel k-step at a time, closed by a scheduling barrier: left alone hipcc hoists the neighbour loads and their
  // fp32 images of ALL k-steps to the top (2 x 64 registers for C_in = 64) and spills.
  bf16x8_t bact[KS_IN][NT];
  const bf16_t* xn = p.xlow + (long)n * p.D * p.H * p.W * p.C_in;
  const float* an = p.ab + (long)n * 2 * p.C_in;
  // neighbours (a, b, d) = low-res voxels (mz-1+a, my-1+b, mx-1+d), clamped (index -1 only feeds face positions)
  int noff[2][2][2];                                          // element offsets inside the sample (< 2^31)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int iz = max(mz - 1 + a, 0), iy = max(my - 1 + b, 0), ix = max(mxc - 1 + d, 0);
        noff[a][b][d] = ((iz * p.H + iy) * p.W + ix) * p.C_in + kb * 8;
      }
#pragma unroll
  for (int ks = 0; ks < KS_IN; ++ks) {
    const int k0 = ks * 32 + kb * 8;
    f32x8_t xin[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (a < pz) continue;                                   // odd output planes read input plane mz only (uniform)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int d = 0; d < 2; ++d)
          xin[a][b][d] = __builtin_convertvector(*reinterpret_cast<const bf16x8_t*>(xn + noff[a][b][d] + ks * 32), f32x8_t);
    }
    float bv1[8], av[8], bv[8];
    VecIO<float, 4>::load(p.b1 + k0, reinterpret_cast<float(&)[4]>(bv1[0]));
    VecIO<float, 4>::load(p.b1 + k0 + 4, reinterpret_cast<float(&)[4]>(bv1[4]));
    VecIO<float, 4>::load(an + k0, reinterpret_cast<float(&)[4]>(av[0]));
    VecIO<float, 4>::load(an + k0 + 4, reinterpret_cast<float(&)[4]>(av[4]));
    VecIO<float, 4>::load(an + p.C_in + k0, reinterpret_cast<float(&)[4]>(bv[0]));
    VecIO<float, 4>::load(an + p.C_in + k0 + 4, reinterpret_cast<float(&)[4]>(bv[4]));
    // Hand-scheduled tap sequence.  A lane's taps are used ONCE each (a tap belongs to one (position, neighbour) pair), so the
    // optimiser, left alone, reads all 18 x 8 tap floats of the k-step from LDS up front and spills them to scratch
    // (215 spilled registers measured).  The order is pinned instead: volatile ds_read_b128 pairs one step ahead of
    // asm-volatile packed FMAs (the scheme of dwconv3d_k3_march_kernel); 2 x 8 tap registers live.
    constexpr int NS = up_steps(pz);
    typedef const volatile __attribute__((address_space(3))) f32x4_t* lds_vol4;
    f32x4_t wq[2][2];
    auto fetch = [&](int st, int buf) {
      const float* wt = wl + up_step(pz, st).tap * p.C_in + k0;
      wq[buf][0] = *(lds_vol4)(wt);
      wq[buf][1] = *(lds_vol4)(wt + 4);
    };
    fetch(0, 0);
    f32x2_t acc[4];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const UpStep u = up_step(pz, st);
      if (u.first) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x2_t{bv1[2 * j], bv1[2 * j + 1]};
      }
      if (st + 1 < NS) fetch(st + 1, (st + 1) & 1);
      const f32x8_t xv = xin[u.a][u.b][u.d];
      const f32x2_t x0{xv[0], xv[1]}, x1{xv[2], xv[3]}, x2{xv[4], xv[5]}, x3{xv[6], xv[7]};
      const f32x4_t wa = wq[st & 1][0], wb = wq[st & 1][1];
      const f32x2_t w0{wa[0], wa[1]}, w1{wa[2], wa[3]}, w2{wb[0], wb[1]}, w3{wb[2], wb[3]};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(x0), "v"(w0));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(x1), "v"(w1));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2]) : "v"(x2), "v"(w2));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[3]) : "v"(x3), "v"(w3));
      if (u.last) {
        const int nt = u.nt, py = nt >> 1, px = nt & 1;
        const bool face = (Pz == 0) | (2 * my + py == 0) | (2 * mxc + px == 0);
        float t8[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { t8[2 * j] = face ? 0.f : acc[j][0]; t8[2 * j + 1] = face ? 0.f : acc[j][1]; }
        // round to bf16 as the un-fused kernel stores t, then the norm affine on the stored value (what pw_mlp reads)
        const f32x8_t tr = __builtin_convertvector(Mma<bf16_t>::from_floats(t8), f32x8_t);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(tr[j], av[j], bv[j]);
        bact[ks][nt] = Mma<bf16_t>::from_floats(v);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- skip rows: requested here (not before the prologue, whose fp32 neighbour images need the registers) so they are in
  //      flight during both GEMMs
  uint4 sk[MO / 2][NT];
  const bf16_t* skn = p.skip + (long)n * (2L * p.D) * Ho * Wo * p.C_out;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pr = 0; pr < MO / 2; ++pr)
      sk[pr][nt] = *reinterpret_cast<const uint4*>(skn + (long)orow[nt] * p.C_out + pr * 32 + kb * 8);

  // ---- GEMM2 accumulators start from the projection bias
  f32x4_t acc2[MO][NT];
#pragma unroll
  for (int pr = 0; pr < MO / 2; ++pr) {
    float b[8];
    VecIO<float, 4>::load(p.b3 + pr * 32 + kb * 8, reinterpret_cast<float(&)[4]>(b[0]));
    VecIO<float, 4>::load(p.b3 + pr * 32 + kb * 8 + 4, reinterpret_cast<float(&)[4]>(b[4]));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc2[2 * pr][nt] = f32x4_t{b[0], b[1], b[2], b[3]};
      acc2[2 * pr + 1][nt] = f32x4_t{b[4], b[5], b[6], b[7]};
    }
  }

  // ---- hidden chunks of 32 units: GEMM1 -> GELU -> GEMM2, all in registers (pw_mlp_kernel's loop)
  for (int hc = 0; hc < p.HC; ++hc) {
    float b2v[8];
    VecIO<float, 4>::load(p.b2 + hc * 32 + kb * 8, reinterpret_cast<float(&)[4]>(b2v[0]));
    VecIO<float, 4>::load(p.b2 + hc * 32 + kb * 8 + 4, reinterpret_cast<float(&)[4]>(b2v[4]));
    f32x4_t acc1[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc1[0][nt] = f32x4_t{b2v[0], b2v[1], b2v[2], b2v[3]};
      acc1[1][nt] = f32x4_t{b2v[4], b2v[5], b2v[6], b2v[7]};
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int ks = 0; ks < KS_IN; ++ks) {
        const bf16x8_t a = p.w2[((long)(hc * 2 + mt) * KS_IN + ks) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc1[mt][nt] = Mma<bf16_t>::mma(a, bact[ks][nt], acc1[mt][nt]);
      }
    }
    bf16x8_t bh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float g[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g[j] = GELU_MODE == 1 ? gelu_fast(acc1[0][nt][j]) : gelu_erf(acc1[0][nt][j]);
        g[4 + j] = GELU_MODE == 1 ? gelu_fast(acc1[1][nt][j]) : gelu_erf(acc1[1][nt][j]);
      }
      bh[nt] = Mma<bf16_t>::from_floats(g);
    }
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
      const bf16x8_t a = p.w3[((long)mo * p.HC + hc) * 64 + lane];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc2[mo][nt] = Mma<bf16_t>::mma(a, bh[nt], acc2[mo][nt]);
    }
  }

  // ---- epilogue (finish_and_store's RES_UPSAMPLE arithmetic with the position known per tile):
  //      face -> skip only; else v + residual + skip, residual = low-res transposed-1x1 result at the all-odd position,
  //      its bias in the stride holes
  if (!live) return;
  bf16_t* yn = p.y + (long)n * (2L * p.D) * Ho * Wo * p.C_out;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int py = nt >> 1, px = nt & 1;
    const bool face = (Pz == 0) | (2 * my + py == 0) | (2 * mx + px == 0);
    const bool odd = pz & py & px;
#pragma unroll
    for (int pr = 0; pr < MO / 2; ++pr) {
      const int o0 = pr * 32 + kb * 8;
      float v[8], s[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc2[2 * pr][nt][j]; v[4 + j] = acc2[2 * pr + 1][nt][j]; }
      VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&sk[pr][nt]), s);
      if (face) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s[j];
      } else {
        float rl[8];
        if (p.res_low && odd) {
          VecIO<bf16_t, 8>::load(p.res_low + ((((long)n * p.D + mz) * p.H + my) * p.W + mx) * p.C_out + o0, rl);
        } else {
          VecIO<float, 4>::load(p.res_bias + o0, reinterpret_cast<float(&)[4]>(rl[0]));
          VecIO<float, 4>::load(p.res_bias + o0 + 4, reinterpret_cast<float(&)[4]>(rl[4]));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] + rl[j] + s[j];
      }
      VecIO<bf16_t, 8>::store(yn + (long)orow[nt] * p.C_out + o0, v);
    }
  }
}

template <int KS_IN, int MO, int GELU_MODE>
__global__ void __launch_bounds__(256, up_waves_per_simd(KS_IN, MO))
pw_mlp_up_kernel(MlpUpParams p) {
  static_assert(MO % 2 == 0, "C_out must be a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) float wl[];          // [27][C_in] taps
  for (int i = threadIdx.x; i < 27 * p.C_in; i += 256) wl[i] = p.taps[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = (int)blockIdx.x * 4 + wave;                // < 2^31 tiles (checked by the host)
  if (tile >= (int)p.tiles) return;
  // tile -> (n, mz, pz, my, x segment): wave-uniform scalar arithmetic
  int tq = tile;
  const int xs = tq % p.XS; tq /= p.XS;
  const int my = tq % p.H; tq /= p.H;
  const int pz = tq & 1; tq >>= 1;
  const int mz = tq % p.D;
  const int n = tq / p.D;
  if (pz) up_tile<KS_IN, MO, GELU_MODE, 1>(p, wl, n, mz, my, xs, lane);
  else up_tile<KS_IN, MO, GELU_MODE, 0>(p, wl, n, mz, my, xs, lane);
}

template <int KS_IN, int MO>
static void launch_up(const MlpUpParams& p, hipStream_t s) {
  dim3 grid((unsigned)((p.tiles + 3) / 4)), block(256);
  const size_t lds = (size_t)27 * p.C_in * sizeof(float);
  if (tuning_get("mlp_exact_gelu", 0)) hipLaunchKernelGGL((pw_mlp_up_kernel<KS_IN, MO, 0>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((pw_mlp_up_kernel<KS_IN, MO, 1>), grid, block, lds, s, p);
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_pw_mlp_up_supported(int C_in, int C_hid, int C_out) {
  if (C_hid % 32 || C_hid < 32) return 0;
  return ((C_in == 64 && C_out == 32) || (C_in == 128 && C_out == 64)) ? 1 : 0;
}

// a->t = the LOW-RES block input x [N][D][H][W][C_in] (not the depthwise output), a->Di/Hi/Wi = the low-res grid,
// a->res = encoder skip at the high resolution, a->res_low / a->res_bias as for PYTC_RES_UPSAMPLE (may be NULL),
// a->y [N][2D][2H][2W][C_out]; taps [27][C_in] fp32 tap-major, dw_bias [C_in] and a->res_bias [C_out] zero vectors when unused.
extern "C" int pytc_pw_mlp_up_fwd(const pytc_mlp_args* a, const float* taps, const float* dw_bias, void* stream) {
  PYTC_REQUIRE(a && a->t && a->ab && a->w2_packed && a->w3_packed && a->b2 && a->b3 && a->y && a->res && taps && dw_bias &&
               a->res_bias, "pw_mlp_up: null pointer (dw_bias and res_bias are zero vectors when unused)");
  PYTC_REQUIRE(a->N >= 1 && a->Di >= 1 && a->Hi >= 1 && a->Wi >= 1, "pw_mlp_up: bad shape");
  if (!pytc_pw_mlp_up_supported(a->C_in, a->C_hid, a->C_out)) {
    set_error("pw_mlp_up: no fused kernel for C_in=%d C_hid=%d C_out=%d", a->C_in, a->C_hid, a->C_out);
    return PYTC_ERR_UNSUPPORTED;
  }
  MlpUpParams p{};
  p.xlow = (const bf16_t*)a->t; p.taps = taps; p.b1 = dw_bias; p.ab = a->ab;
  p.w2 = (const bf16x8_t*)a->w2_packed; p.b2 = a->b2; p.w3 = (const bf16x8_t*)a->w3_packed; p.b3 = a->b3;
  p.skip = (const bf16_t*)a->res; p.res_low = (const bf16_t*)a->res_low; p.res_bias = a->res_bias; p.y = (bf16_t*)a->y;
  p.N = a->N; p.D = a->Di; p.H = a->Hi; p.W = a->Wi; p.XS = (a->Wi + 15) / 16;
  p.C_in = a->C_in; p.C_hid = a->C_hid; p.C_out = a->C_out; p.HC = a->C_hid / 32;
  p.tiles = (long)a->N * a->Di * 2 * a->Hi * p.XS;
  PYTC_REQUIRE(p.tiles < (1L << 31) - 8, "pw_mlp_up: too many tiles");
  hipStream_t s = (hipStream_t)stream;
  if (a->C_in == 64) launch_up<2, 2>(p, s);
  else launch_up<4, 4>(p, s);
  PYTC_LAUNCH_CHECK("pw_mlp_up");
  return PYTC_OK;
}
