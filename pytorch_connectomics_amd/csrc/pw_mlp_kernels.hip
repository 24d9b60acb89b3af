// Fused MedNeXt channel mixer (bf16):   y = W3 * gelu(W2 * (a*t + b) + b2) + b3  (+ residual variants)
// One launch replaces GroupNorm-apply, 1x1 expand, GELU, 1x1 project and the residual add; the expanded
// r*C tensor never leaves the registers.
//
// Both GEMMs run in the transposed form of pw_kernels.hip (weights = MFMA A operand, voxels = N), so
//   * the normalised input tile is loaded straight from NDHWC memory into B-operand registers (16 B/lane);
//   * with the "paired row" weight packing (pw_common.h) the fp32 accumulator of GEMM1, after GELU and a
//     v_cvt_pk_bf16_f32, IS the B operand of GEMM2 in natural k order -- no LDS, no cross-lane shuffles;
//   * every lane ends with 8 consecutive output channels of one voxel -> one 16-byte store, and reads its
//     residual the same way.
// HBM traffic per voxel = C_in (t) + C_out (residual) + C_out (y) elements: the byte floor of the op.
#include <cmath>
#include "pw_common.h"

namespace pytc {

struct MlpParams {
  const bf16_t* t;
  const float* ab;
  const bf16x8_t* w2;
  const float* b2;
  const bf16x8_t* w3;
  const float* b3;
  EpiParams e;
  long rps;        // rows per sample (input == output rows)
  int C_in, C_hid, C_out, HC;   // HC = C_hid / 32
  int w3_f16;      // w3 is the fp16 paired image: packed-fp16 GELU + f16 MFMA for the projection (GELU_MODE 3)
  // per-sample expand operands (ab == nullptr): the GroupNorm affine is folded into the expanding conv by
  // groupnorm_fold_mlp_kernel -- W2_n = W2 * diag(a_n), b2_n = b2 + W2 * b_n -- so the B operand of GEMM1 is the RAW bf16 tile
  // (no unpack / fma / repack per element: the mixers are bound by VALU issue); w2 / b2 then hold N
  // images / vectors back to back
  long w2_stride;  // bf16x8 elements per sample image
  // fused output head (HEAD kernels): logits[o] = head_b[o] + sum_c head_w[o][c] * bf16(y[c]),  o < n_head <= 16
  const bf16x8_t* head_w;   // A fragment image [64 lanes][8]: lane (r, kb) holds head[o = r][c = kb*8 .. +7], bf16
  const float* head_b;
  float* head_y;   // [N][rps][n_head] fp32
  int n_head, store_y;
  // PROJ form of pw_mlp_dma_kernel: a 32 -> 32 conv of the block output in the epilogue (head_w = its paired bf16 image, 2 tiles x 64 lanes x 8;
  // head_b = its bias [32]); z = bf16(W bf16(y) + b) goes to proj_y [N][rps][32], y itself only when store_y
  bf16_t* proj_y;
  // STEMRES kernels: the block's residual is the stem output, recomputed from the 1-channel network input
  //   res[c] = bf16(stem_w[c] * stem_x[voxel] + stem_b[c])    (what the un-fused stem kernel would have stored)
  const float* stem_x;     // [N][rps] fp32
  const float* stem_w;
  const float* stem_b;
  // STOREH kernels (training forward): the pre-activation of the hidden layer is also written, [N][rps][C_hid] bf16
  // (round_h with hp == nullptr: rounded to bf16 exactly as the storing form rounds it, but not written -- the backward of the
  // full-resolution blocks rebuilds it, mixer_bwd_rc_kernel)
  bf16_t* hp;
  int round_h;
  // BWD kernels (training backward, data gradients of both GEMMs in one launch): the "activation" between the GEMMs is
  // the multiplication by GELU'(hp_in), and hp receives the product (d loss / d pre-activation) for the weight gradient
  const bf16_t* hp_in;
};

// GELU by table: the mixer is VALU bound on its activation (SQ counters of 64->128->32: 70 % VALU busy, v_exp_f32 and
// v_rcp_f32 are about 60 % of it).  1024 segments of [-8, 8): y = a_i + b_i * x with (a_i, b_i) from the exact erf GELU
// in double precision -> |error| <= 2.4e-5 (h^2/8 * max f''), the class of `gelu_fast`; outside the range the end
// segments extend linearly (slopes 0 and 1 to 1e-14).  One fma + clamp + cvt + ds_read_b64 + fma per element.
constexpr int GELU_LUT_N = 1024;
__device__ float2 g_gelu_lut[GELU_LUT_N];

__device__ __forceinline__ float gelu_lut(const float2* __restrict__ tab, float x) {
  const float t = fminf(fmaxf(fmaf(x, 64.0f, 512.0f), 0.0f), (float)(GELU_LUT_N - 1));
  const float2 ab = tab[(int)t];
  return fmaf(ab.y, x, ab.x);
}

// Occupancy hint: with a bare __launch_bounds__(256) hipcc budgets for one wave per SIMD and spends registers freely
// (176 VGPRs for 128->256->64, i.e. 2 waves/SIMD); the waves of this kernel spend > 40 % of their life waiting on
// loads (SQ_WAIT_ANY), so the kernels are compiled for the occupancy their live state allows (no spills).
constexpr int mlp_waves_per_simd(int ks, int mo, int nt) {
  if (ks >= 8 && nt == 1) return 1;                             // deep levels: weights-in-flight need the registers
  const int regs = ks * nt * 4 + mo * nt * 4 + 2 * nt * 4;     // operand tile + both accumulators
  return regs <= 64 ? 4 : (regs <= 112 ? 3 : (regs <= 200 ? 2 : 1));
}

// GELU_MODE: 0 = erf (A&S 7.1.26), 1 = sigmoid-form minimax (gelu_fast), 2 = table, 3 = packed fp16 polynomial (gelu_h2;
// the projection then runs on v_mfma_f32_16x16x32_f16 with the fp16 image of W3)
// HEAD: the network's 1x1x1 output projection rides in the epilogue of the LAST mixer (C_out = 32): the block output is
// rounded to bf16 exactly as the un-fused path stores it, is itself the B fragment of one more 16x16x32 MFMA against the head
// weights (rows beyond n_head are zero), whose result lanes write the fp32 logits; the 64 B / voxel of
// block output are not written at all when nothing else reads them (store_y = 0).
// STOREH: training forward -- the same kernel also stores the hidden pre-activation (GEMM1 + bias, bf16) that the backward
// pass needs (GELU' and the weight gradient of the projection), so the two-GEMM training forward is one launch and the
// hidden tensor is written once and not read back in the forward.
// BWD: dX = W2^T ((W3^T dY) * GELU'(hp)) -- the two data-gradient GEMMs of a block's mixer with the derivative of the
// activation between them: t := dY, w2 := W3^T, w3 := W2^T, no affine, no biases, no residual; the intermediate
// (d loss / d hp, bf16) is stored to p.hp for the weight gradient of the expanding conv.
template <int KS_IN, int MO, int NT, int GELU_MODE, bool HEAD = false, bool STEMRES = false, bool STOREH = false, bool BWD = false>
__global__ void __launch_bounds__(256, mlp_waves_per_simd(KS_IN, MO, NT))
pw_mlp_kernel(MlpParams p) {
  // channel counts are the template's: row offsets compile to shifts (64-bit multiplies by a runtime C are quarter-rate instructions)
  constexpr int CIN = KS_IN * 32, COUT = MO * 16;
  static_assert(MO % 2 == 0, "C_out must be a multiple of 32");
  static_assert(!STEMRES || (MO == 2 && (MO / 2) * NT <= 4), "the stem-recomputing residual covers C_out = 32");
  static_assert(!HEAD || MO == 2, "the fused head covers C_out = 32");
  __shared__ __attribute__((aligned(16))) float2 lut[GELU_MODE == 2 ? GELU_LUT_N : 1];
  if constexpr (GELU_MODE == 2) {
    const uint4* src = reinterpret_cast<const uint4*>(g_gelu_lut);
    uint4* dst = reinterpret_cast<uint4*>(lut);
    dst[threadIdx.x] = src[threadIdx.x];
    dst[threadIdx.x + 256] = src[threadIdx.x + 256];
    __syncthreads();
  }
  constexpr bool PREFETCH_RES = (MO / 2) * NT <= 4;   // residual rows ride along with the input loads
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.y;
  constexpr bool WARM = KS_IN * MO >= 32;    // >= 64 KB of weights: warm this XCD's L2 (see warm_l2)
  constexpr int WPER = KS_IN * MO >= 512 ? 6 : 2;   // few workgroups at the deepest level: more lines per lane
  WarmRegs<WPER> warm;
  const bool folded = p.ab == nullptr;        // wave-uniform: per-sample expand operands (see MlpParams)
  const bf16x8_t* w2 = p.w2 + (folded ? (long)n * p.w2_stride : 0L);
  const float* b2 = p.b2 + (folded ? (long)n * p.C_hid : 0L);
  if (WARM) {
    warm_l2(w2, (long)p.C_hid * CIN * 2, warm, 0);
    warm_l2(p.w3, (long)COUT * p.C_hid * 2, warm, WPER);
  }
  const long row0 = ((long)blockIdx.x * 4 + wave) * (NT * 16);
  if (row0 >= p.rps) {
    // a wave must not end with warm-up loads in flight: their data would land in registers of whichever wave is
    // allocated next
    if (WARM) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    return;
  }
  const int r = lane & 15, kb = lane >> 4;

  long orow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) orow[nt] = row0 + nt * 16 + r;

  // ---- B operand of GEMM1: normalised input, 8 consecutive channels per lane per k-step
  bf16x8_t bact[KS_IN][NT];
  const bf16_t* tn = p.t + (long)n * p.rps * CIN;
  if (folded) {
#pragma unroll
    for (int ks = 0; ks < KS_IN; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const long rr = orow[nt] < p.rps ? orow[nt] : p.rps - 1;
        bact[ks][nt] = ld_stream(reinterpret_cast<const bf16x8_t*>(tn + rr * CIN + ks * 32 + kb * 8), p.e.nt);
      }
  }
  const float* an = p.ab + (long)n * 2 * CIN;
#pragma unroll
  for (int ks = 0; ks < KS_IN; ++ks) {
    if (folded) break;
    const int k0 = ks * 32 + kb * 8;
    float av[8], bv[8];
    VecIO<float, 4>::load(an + k0, reinterpret_cast<float(&)[4]>(av[0]));
    VecIO<float, 4>::load(an + k0 + 4, reinterpret_cast<float(&)[4]>(av[4]));
    VecIO<float, 4>::load(an + CIN + k0, reinterpret_cast<float(&)[4]>(bv[0]));
    VecIO<float, 4>::load(an + CIN + k0 + 4, reinterpret_cast<float(&)[4]>(bv[4]));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const long rr = orow[nt] < p.rps ? orow[nt] : p.rps - 1;
      float v[8];
      VecIO<bf16_t, 8>::load(tn + rr * CIN + k0, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], av[j], bv[j]);
      bact[ks][nt] = Mma<bf16_t>::from_floats(v);
    }
  }

  if (WARM) warm_l2_done(warm);              // bact is built: the (older) warm-up loads have landed

  // ---- residual / skip rows: issue their loads now so they are in flight during both GEMMs
  uint4 rpre[PREFETCH_RES ? MO / 2 : 1][PREFETCH_RES ? NT : 1];
  const bool use_pre = PREFETCH_RES && p.e.res_mode != PYTC_RES_NONE;
  if constexpr (STEMRES) {
    const float* sx = p.stem_x + (long)n * p.rps;
    float sw[8], sb[8];
    VecIO<float, 4>::load(p.stem_w + kb * 8, reinterpret_cast<float(&)[4]>(sw[0]));
    VecIO<float, 4>::load(p.stem_w + kb * 8 + 4, reinterpret_cast<float(&)[4]>(sw[4]));
    VecIO<float, 4>::load(p.stem_b + kb * 8, reinterpret_cast<float(&)[4]>(sb[0]));
    VecIO<float, 4>::load(p.stem_b + kb * 8 + 4, reinterpret_cast<float(&)[4]>(sb[4]));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float xv = sx[orow[nt] < p.rps ? orow[nt] : p.rps - 1];
      float rv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[j] = fmaf(sw[j], xv, sb[j]);
      rpre[0][nt] = __builtin_bit_cast(uint4, Mma<bf16_t>::from_floats(rv));
    }
  } else if (use_pre) {
    const bf16_t* resn = reinterpret_cast<const bf16_t*>(p.e.res) + (long)n * p.rps * COUT;
#pragma unroll
    for (int pr = 0; pr < (PREFETCH_RES ? MO / 2 : 1); ++pr)
#pragma unroll
      for (int nt = 0; nt < (PREFETCH_RES ? NT : 1); ++nt) {
        const long rr = orow[nt] < p.rps ? orow[nt] : p.rps - 1;
        rpre[pr][nt] = ld_stream(reinterpret_cast<const uint4*>(resn + rr * COUT + pr * 32 + kb * 8), p.e.nt);
      }
  }

  // ---- GEMM2 accumulators start from the projection bias (8 consecutive channels per lane per pair)
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

  // ---- loop over hidden chunks of 32 units: GEMM1 -> GELU -> GEMM2, all in registers
  for (int hc = 0; hc < p.HC; ++hc) {
    float b2v[8];
    VecIO<float, 4>::load(b2 + hc * 32 + kb * 8, reinterpret_cast<float(&)[4]>(b2v[0]));
    VecIO<float, 4>::load(b2 + hc * 32 + kb * 8 + 4, reinterpret_cast<float(&)[4]>(b2v[4]));
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
        const bf16x8_t a = w2[((long)(hc * 2 + mt) * KS_IN + ks) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc1[mt][nt] = Mma<bf16_t>::mma(a, bact[ks][nt], acc1[mt][nt]);
      }
    }
    bf16x8_t bh[NT];
    h8_t bhh[GELU_MODE == 3 ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float g[8];
      if constexpr (BWD) {
        const long rr = orow[nt] < p.rps ? orow[nt] : p.rps - 1;
        float hv[8];
        VecIO<bf16_t, 8>::load(p.hp_in + ((long)n * p.rps + rr) * p.C_hid + hc * 32 + kb * 8, hv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[j] = acc1[0][nt][j] * gelu_erf_grad(hv[j]);
          g[4 + j] = acc1[1][nt][j] * gelu_erf_grad(hv[4 + j]);
        }
        bh[nt] = Mma<bf16_t>::from_floats(g);
        if (orow[nt] < p.rps)
          *reinterpret_cast<bf16x8_t*>(p.hp + ((long)n * p.rps + orow[nt]) * p.C_hid + hc * 32 + kb * 8) = bh[nt];
        continue;
      }
      if constexpr (STOREH) {
        // the backward pass evaluates GELU' and GELU at the STORED (bf16) pre-activation: use the same value here
        float pre[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { pre[j] = acc1[0][nt][j]; pre[4 + j] = acc1[1][nt][j]; }
        const bf16x8_t hb = Mma<bf16_t>::from_floats(pre);
        if (p.hp && orow[nt] < p.rps)
          *reinterpret_cast<bf16x8_t*>(p.hp + ((long)n * p.rps + orow[nt]) * p.C_hid + hc * 32 + kb * 8) = hb;
        const f32x8_t hr = __builtin_convertvector(hb, f32x8_t);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[0][nt][j] = hr[j]; acc1[1][nt][j] = hr[4 + j]; }
      }
      if constexpr (GELU_MODE == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = acc1[0][nt][j]; g[4 + j] = acc1[1][nt][j]; }
        bhh[nt] = gelu_h8_from_f32(g);
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g[j] = GELU_MODE == 2 ? gelu_lut(lut, acc1[0][nt][j]) : (GELU_MODE == 1 ? gelu_fast(acc1[0][nt][j]) : gelu_erf(acc1[0][nt][j]));
        g[4 + j] = GELU_MODE == 2 ? gelu_lut(lut, acc1[1][nt][j]) : (GELU_MODE == 1 ? gelu_fast(acc1[1][nt][j]) : gelu_erf(acc1[1][nt][j]));
      }
      bh[nt] = Mma<bf16_t>::from_floats(g);
    }
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
      if constexpr (GELU_MODE == 3) {
        const h8_t a = reinterpret_cast<const h8_t*>(p.w3)[((long)mo * p.HC + hc) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[mo][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhh[nt], acc2[mo][nt], 0, 0, 0);
      } else {
        const bf16x8_t a = p.w3[((long)mo * p.HC + hc) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[mo][nt] = Mma<bf16_t>::mma(a, bh[nt], acc2[mo][nt]);
      }
    }
  }

  if constexpr (HEAD) {
    bf16_t* yn = reinterpret_cast<bf16_t*>(p.e.y) + (long)n * p.rps * 32;
    const bf16_t* resn = p.e.res ? reinterpret_cast<const bf16_t*>(p.e.res) + (long)n * p.rps * 32 : nullptr;
    float* hy = p.head_y + (long)n * p.rps * p.n_head;
    const bf16x8_t ah = p.head_w[lane];        // A fragment of the head: row r = output o (zero rows beyond n_head)
    float hb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hb[i] = (p.head_b && kb * 4 + i < p.n_head) ? p.head_b[kb * 4 + i] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc2[0][nt][j]; v[4 + j] = acc2[1][nt][j]; }
      const bool live = orow[nt] < p.rps;
      if (p.e.res_mode == PYTC_RES_ADD) {
        float rv[8];
        if (use_pre) VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rpre[0][PREFETCH_RES ? nt : 0]), rv);
        else VecIO<bf16_t, 8>::load(resn + (live ? orow[nt] : p.rps - 1) * 32 + kb * 8, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
      }
      // this lane's 8 output channels of voxel r, rounded as the un-fused path stores them = the B fragment of
      // logits^T[o][voxel] = sum_c head[o][c] * y[c][voxel]: one more MFMA, no cross-lane traffic
      const bf16x8_t ob = Mma<bf16_t>::from_floats(v);
      if (p.store_y && live) *reinterpret_cast<bf16x8_t*>(yn + orow[nt] * 32 + kb * 8) = ob;
      const f32x4_t h = Mma<bf16_t>::mma(ah, ob, f32x4_t{0.f, 0.f, 0.f, 0.f});
      if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kb * 4 + i < p.n_head) hy[orow[nt] * p.n_head + kb * 4 + i] = h[i] + hb[i];
      }
    }
    return;
  }
  // ---- epilogue: 8 consecutive channels per lane per tile pair
  // RES_UPSAMPLE needs each row's position in the output grid: one division pair per WAVE (its first row, wave-uniform), then
  // per lane an add and a carry per tile -- not two divisions per stored 16 bytes
  int upos[NT][3];
  const bool ups = p.e.res_mode == PYTC_RES_UPSAMPLE;
  if (ups) {
    const unsigned ur = (unsigned)row0, gw = (unsigned)p.e.Go_w, gh = (unsigned)p.e.Go_h;
    const unsigned t0 = ur / gw;
    const int bx = (int)(ur - t0 * gw);
    const int bz = (int)(t0 / gh);
    const int by = (int)(t0 - (unsigned)bz * gh);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int px = bx + nt * 16 + r, py = by, pz = bz;
      while (px >= p.e.Go_w) { px -= p.e.Go_w; ++py; }
      while (py >= p.e.Go_h) { py -= p.e.Go_h; ++pz; }
      upos[nt][0] = pz; upos[nt][1] = py; upos[nt][2] = px;
    }
  }
#pragma unroll
  for (int pr = 0; pr < MO / 2; ++pr) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (orow[nt] >= p.rps) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc2[2 * pr][nt][j];
        v[4 + j] = acc2[2 * pr + 1][nt][j];
      }
      if (use_pre) {
        float pre[8];
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rpre[PREFETCH_RES ? pr : 0][PREFETCH_RES ? nt : 0]), pre);
        finish_and_store<bf16_t, 8, false, COUT>(v, p.e, n, orow[nt], pr * 32 + kb * 8, pre, ups ? upos[nt] : nullptr);
      } else {
        finish_and_store<bf16_t, 8, false, COUT>(v, p.e, n, orow[nt], pr * 32 + kb * 8, nullptr, ups ? upos[nt] : nullptr);
      }
    }
  }
}

// Measured and removed (round 4, profiles/r04_mixer_occupancy_and_issue_rates.txt): the level-0 / level-1 shapes compiled for 4 / 5 /
// 6 waves per SIMD (launch-bounds override, NT = 2 and 4): no shape gets faster, the spilling ones get slower (64->128->32: 910 ->
// 1106 us at 4 waves with 36 spilled registers) -- occupancy is not what these kernels lack.
// Also measured and removed (round 2, profiles/r02_mixer_split_and_lds_weights.txt): (a) the four waves of a workgroup sharing
// one voxel tile and splitting the hidden chunks between them (fixed-order LDS reduction): wins only at 7^3, loses at 14^3;
// (b) each hidden chunk's weight fragments fetched once per workgroup into double-buffered LDS instead of streamed from L2 by
// every wave: bit-identical and slower at every shape -- the per-chunk barrier exposes the next chunk's load latency that
// independent waves otherwise hide.  The mixers are not limited by L2 weight bandwidth.
// Measured and removed (round 2, profiles/r02_mixer_two_tiles_per_wave.txt): a variant in which a wave owns TWO voxel tiles and
// requests the rows of both before computing the first (to overlap one tile's loads with the other's GEMMs inside a wave) was
// bit-identical and SLOWER at every shape (forward 9.39 -> 10.1 ms; 128->256->128: 0.26 -> 0.61 ms): the registers of the second
// tile cost a wave per SIMD, and this kernel lives on occupancy -- its per-wave critical path (MFMA -> GELU -> MFMA dependency
// chains) is hidden by other waves, not by memory-level parallelism inside one.

// ---- level-0 mixer with its operand rows landing in LDS by DMA, one tile ahead (round 5) ------------------------------------------------------
// pw_mlp_kernel<1, 2, 4, 3> is memory bound at ~5.0 TB/s where a bare read-read-write kernel reaches 5.9 (profiles/r05_stream_policy.txt): its waves
// hold 8 KB of loads in flight only until they start computing (60 % of their cycles waiting), ~77 KB per CU on average against the ~100 KB the
// loaded memory system's 4 us want.  Prefetching the next tile into REGISTERS costs a wave per SIMD and lost twice (rounds 2 and 4).  Here the next
// tile's rows travel by `global_load_lds_dwordx4` into a wave-private LDS landing area (8 KB per wave: each lane's 16 bytes land at its own
// lane slot, so reading them back is a conflict-free identity map) while the wave computes the current tile from registers: no register cost, 12
// waves per CU instead of 16 (LDS: 41 - 51 KB per workgroup with the operands below), and every wave has a tile in flight all the time.  A wave
// walks tiles gw, gw + W, gw + 2W ... of its sample (W = waves of the launch: neighbouring waves stay on neighbouring rows).  Order inside an
// iteration: wait (the DMA of this tile, and the stores of the tile before last, have had a whole compute phase) -> stores of the PREVIOUS tile's
// results -> landing area to registers -> DMA of the next tile -> compute.  (Loads and stores share vmcnt and return out of order with respect
// to each other: the stores must not sit between a DMA and its wait.)
// Same MFMA order, GELU and epilogue arithmetic as pw_mlp_kernel<1, 2, 4, 3> with per-sample operands: bit-identical output.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {      // lane l's dword lands at lds_dst + 4 l
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// KIND 0: plain / residual-add epilogue; 1: the first block's residual rows recomputed from the raw input (pw_mlp_kernel's STEMRES: the row's
// fp32 input voxel rides the DMA as a dword); 2: the last block with the output heads fused (pw_mlp_kernel's HEAD); 3: the last block with a
// 32 -> 32 conv of its output fused (the input projection of merged task heads): the rounded block output is the B operand of two more MFMAs
template <int HCT, int KIND>
__global__ void __launch_bounds__(256, 3)
pw_mlp_dma_kernel(MlpParams p, int waves_per_sample) {
  constexpr int NT = 4, CIN = 32, COUT = 32;
  constexpr bool STEM = KIND == 1, HEAD = KIND == 2, PROJ = KIND == 3;
  // 32 KB of landing area + the sample's operands (the compiler's own loads of them inside the loop would share vmcnt with the DMA and, returning
  // in order behind it, wait for the prefetch they are meant to overlap): 40.4 KB at HCT = 2, 49.4 KB at 4 -> 3 workgroups per CU
  __shared__ __attribute__((aligned(16))) uint4 land[4][2][NT * 64];        // [wave][t | residual (STEM: input voxels)][tile][lane]
  __shared__ __attribute__((aligned(16))) uint4 w2s[HCT * 2 * 64], w3s[2 * HCT * 64];
  __shared__ __attribute__((aligned(16))) float b2s[HCT * 32], b3s[32];
  __shared__ __attribute__((aligned(16))) float auxa[STEM ? 32 : (HEAD ? 256 : (PROJ ? 512 : 4))], auxb[(STEM || PROJ) ? 32 : (HEAD ? 16 : 4)];   // stem w, b / head (projection) image, bias
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.y;
  const int r = lane & 15, kb = lane >> 4;
  const long tiles = (p.rps + NT * 16 - 1) / (NT * 16);
  const bool with_res = STEM || p.e.res_mode == PYTC_RES_ADD;
  const bf16_t* tn = p.t + (long)n * p.rps * CIN;
  const bf16_t* resn = (!STEM && with_res) ? reinterpret_cast<const bf16_t*>(p.e.res) + (long)n * p.rps * COUT : nullptr;
  const float* sx = STEM ? p.stem_x + (long)n * p.rps : nullptr;
  bf16_t* yn = reinterpret_cast<bf16_t*>(p.e.y) + (long)n * p.rps * COUT;
  float* hy = HEAD ? p.head_y + (long)n * p.rps * p.n_head : nullptr;
  bf16_t* zn = PROJ ? p.proj_y + (long)n * p.rps * 32 : nullptr;
  {
    const uint4* w2 = reinterpret_cast<const uint4*>(p.w2 + (long)n * p.w2_stride);
    const uint4* w3 = reinterpret_cast<const uint4*>(p.w3);
    for (int i = threadIdx.x; i < HCT * 2 * 64; i += 256) { w2s[i] = w2[i]; w3s[i] = w3[i]; }
    if (threadIdx.x < HCT * 32) b2s[threadIdx.x] = p.b2[(long)n * p.C_hid + threadIdx.x];
    if (threadIdx.x < 32) b3s[threadIdx.x] = p.b3[threadIdx.x];
    if constexpr (STEM) {
      if (threadIdx.x < 32) { auxa[threadIdx.x] = p.stem_w[threadIdx.x]; auxb[threadIdx.x] = p.stem_b[threadIdx.x]; }
    }
    if constexpr (HEAD) {
      reinterpret_cast<float*>(auxa)[threadIdx.x] = reinterpret_cast<const float*>(p.head_w)[threadIdx.x];     // 64 lanes x 16 bytes
      if (threadIdx.x < 16) auxb[threadIdx.x] = (p.head_b && (int)threadIdx.x < p.n_head) ? p.head_b[threadIdx.x] : 0.f;
    }
    if constexpr (PROJ) {
      auxa[threadIdx.x] = reinterpret_cast<const float*>(p.head_w)[threadIdx.x];                                 // 2 tiles x 64 lanes x 16 bytes
      auxa[threadIdx.x + 256] = reinterpret_cast<const float*>(p.head_w)[threadIdx.x + 256];
      if (threadIdx.x < 32) auxb[threadIdx.x] = p.head_b ? p.head_b[threadIdx.x] : 0.f;
    }
  }
  __syncthreads();
  // the compiler's own loads above are awaited before the first DMA is issued (its waits do not know about asm-issued loads)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&land[wave][0][0]);
  auto request = [&](long tile) {                  // 4 (+ 4) DMA loads: lane (r, kb) of tile nt fetches row tile*64 + nt*16 + r, piece kb
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      long row = tile * (NT * 16) + nt * 16 + r;
      row = row < p.rps ? row : p.rps - 1;
      glds16(tn + row * CIN + kb * 8, lbase + (unsigned)(nt * 64) * 16);
      if constexpr (STEM) glds4(sx + row, lbase + (unsigned)(NT * 64) * 16 + (unsigned)(nt * 64) * 4);
      else if (with_res) glds16(resn + row * COUT + kb * 8, lbase + (unsigned)((NT + nt) * 64) * 16);
    }
  };
  bf16x8_t out[NT];                                // results of the previous tile (stored at the start of the next iteration)
  f32x4_t hout[HEAD ? NT : 1];
  bf16x8_t zout[PROJ ? NT : 1];
  auto store = [&](long tile) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const long row = tile * (NT * 16) + nt * 16 + r;
      if (row >= p.rps) continue;
      if (!(HEAD || PROJ) || p.store_y) *reinterpret_cast<bf16x8_t*>(yn + row * COUT + kb * 8) = out[nt];
      if constexpr (PROJ) *reinterpret_cast<bf16x8_t*>(zn + row * 32 + kb * 8) = zout[nt];
      if constexpr (HEAD) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kb * 4 + i < p.n_head) hy[row * p.n_head + kb * 4 + i] = hout[nt][i];
      }
    }
  };
  const long gw = (long)blockIdx.x * 4 + wave;
  long tile = gw;
  if (tile >= tiles) return;
  request(tile);
  long out_tile = -1;
  for (; tile < tiles; tile += waves_per_sample) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (out_tile >= 0) store(out_tile);
    bf16x8_t bact[NT];
    uint4 rpre[STEM ? 1 : NT];
    float xin[STEM ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bact[nt] = __builtin_bit_cast(bf16x8_t, land[wave][0][nt * 64 + lane]);
      if constexpr (STEM) xin[nt] = reinterpret_cast<const float*>(&land[wave][1][0])[nt * 64 + lane];
      else if (with_res) rpre[nt] = land[wave][1][nt * 64 + lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the landing area is free again before the next DMA may write it
    __builtin_amdgcn_wave_barrier();
    if (tile + waves_per_sample < tiles) request(tile + waves_per_sample);
    // ---- the mixer of pw_mlp_kernel<1, 2, 4, 3> on this tile
    f32x4_t acc2[2][NT];
    {
      const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(&b3s[kb * 8]), hi = *reinterpret_cast<const f32x4_t*>(&b3s[kb * 8 + 4]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { acc2[0][nt] = lo; acc2[1][nt] = hi; }
    }
#pragma unroll
    for (int hc = 0; hc < HCT; ++hc) {
      f32x4_t acc1[2][NT];
      {
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(&b2s[hc * 32 + kb * 8]), hi = *reinterpret_cast<const f32x4_t*>(&b2s[hc * 32 + kb * 8 + 4]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc1[0][nt] = lo; acc1[1][nt] = hi; }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bf16x8_t a2 = __builtin_bit_cast(bf16x8_t, w2s[(hc * 2 + mt) * 64 + lane]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc1[mt][nt] = Mma<bf16_t>::mma(a2, bact[nt], acc1[mt][nt]);
      }
      h8_t bhh[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float g[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = acc1[0][nt][j]; g[4 + j] = acc1[1][nt][j]; }
        bhh[nt] = gelu_h8_from_f32(g);
      }
#pragma unroll
      for (int mo = 0; mo < 2; ++mo) {
        const h8_t a3 = __builtin_bit_cast(h8_t, w3s[(mo * HCT + hc) * 64 + lane]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[mo][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, bhh[nt], acc2[mo][nt], 0, 0, 0);
      }
    }
    float sw[STEM ? 8 : 1], sb[STEM ? 8 : 1];
    if constexpr (STEM) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { sw[j] = auxa[kb * 8 + j]; sb[j] = auxb[kb * 8 + j]; }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc2[0][nt][j]; v[4 + j] = acc2[1][nt][j]; }
      if constexpr (STEM) {
        float rv[8], pre[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) rv[j] = fmaf(sw[j], xin[nt], sb[j]);
        const bf16x8_t rb = Mma<bf16_t>::from_floats(rv);          // the residual row as the stem would have stored it
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rb), pre);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += pre[j];
      } else if (with_res) {
        float pre[8];
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rpre[nt]), pre);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += pre[j];
      }
      out[nt] = Mma<bf16_t>::from_floats(v);
      if constexpr (HEAD) {
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, reinterpret_cast<const uint4*>(auxa)[lane]);
        const f32x4_t hb = *reinterpret_cast<const f32x4_t*>(&auxb[kb * 4]);
        const f32x4_t h = Mma<bf16_t>::mma(ah, out[nt], f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int i = 0; i < 4; ++i) hout[nt][i] = h[i] + hb[i];
      }
      if constexpr (PROJ) {
        // z = W bf16(y) + b: accumulators start from the bias, one K = 32 MFMA per 16-channel tile, 8 consecutive channels per lane (paired rows)
        f32x4_t z0 = *reinterpret_cast<const f32x4_t*>(&auxb[kb * 8]), z1 = *reinterpret_cast<const f32x4_t*>(&auxb[kb * 8 + 4]);
        z0 = Mma<bf16_t>::mma(__builtin_bit_cast(bf16x8_t, reinterpret_cast<const uint4*>(auxa)[lane]), out[nt], z0);
        z1 = Mma<bf16_t>::mma(__builtin_bit_cast(bf16x8_t, reinterpret_cast<const uint4*>(auxa)[64 + lane]), out[nt], z1);
        float zv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { zv[j] = z0[j]; zv[4 + j] = z1[j]; }
        zout[nt] = Mma<bf16_t>::from_floats(zv);
      }
    }
    out_tile = tile;
  }
  store(out_tile);
}

// which launches take the DMA form: level-0 width, per-sample operands, fp16 projection image, a launch of at least `mlp_dma_rows` rows
static bool mlp_dma_applies(const pytc_mlp_args* a, const MlpParams& p) {
  return a->per_sample && p.w3_f16 && a->C_in == 32 && a->C_out == 32 && (p.HC == 2 || p.HC == 3 || p.HC == 4) &&
         (a->res_mode == PYTC_RES_NONE || a->res_mode == PYTC_RES_ADD) && tuning_get("mlp_dma", 1) != 0 &&
         (long)a->N * a->rows_per_sample >= (long)tuning_get("mlp_dma_rows", 1 << 20);
}
template <int KIND>
static void mlp_dma_launch(const pytc_mlp_args* a, const MlpParams& p, hipStream_t s) {
  const long tiles = (a->rows_per_sample + 63) / 64;
  // workgroups of the launch per sample: several waves' worth per CU slot (3 workgroups per CU by LDS); every wave gets >= 2 tiles
  long wgs = ((long)tuning_get("mlp_dma_grid", 3072) + a->N - 1) / a->N;
  const long most = (tiles / 2 + 3) / 4;
  if (wgs > most) wgs = most < 1 ? 1 : most;
  if (const int forced = tuning_get("mlp_dma_wgs", 0); forced > 0) wgs = forced;
  dim3 grid((unsigned)wgs, (unsigned)a->N), block(256);
  const int wps = (int)(wgs * 4);
  if (p.HC == 2) hipLaunchKernelGGL((pw_mlp_dma_kernel<2, KIND>), grid, block, 0, s, p, wps);
  else if (p.HC == 3) hipLaunchKernelGGL((pw_mlp_dma_kernel<3, KIND>), grid, block, 0, s, p, wps);
  else hipLaunchKernelGGL((pw_mlp_dma_kernel<4, KIND>), grid, block, 0, s, p, wps);
}

template <typename OUT>
__global__ void __launch_bounds__(256)
pw_pack_paired_kernel(const float* __restrict__ w, int C_out, int C_in, int transposed,
                      OUT* __restrict__ packed, int KG, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int j = (int)(i % 8);
  long t = i / 8;
  int lane = (int)(t % 64);
  t /= 64;
  int kg = (int)(t % KG);
  int T = (int)(t / KG);
  int o = paired_row(T, lane & 15);
  int k = kg * 32 + (lane >> 4) * 8 + j;
  float v = 0.f;
  if (o < C_out && k < C_in) v = transposed ? w[(long)k * C_out + o] : w[(long)o * C_in + k];
  if constexpr (sizeof(OUT) == 2 && !__is_same(OUT, bf16_t)) packed[i] = (OUT)v;      // fp16 image (round to nearest even)
  else packed[i] = from_f32<bf16_t>(v);
}

// Every per-step weight re-layout of a model in ONE launch (training: the weights change at every optimizer step, so the
// MFMA images of all 1x1x1 convs -- forward and transposed for the data gradients -- and the tap-major depthwise stencils
// have to be rebuilt once per step; as ~120 separate launches of 5 us kernels they cost 0.8 ms of a 36 ms MedNeXt-S step).
// table [n_items][8] int64: {src fp32, dst, kind, C_out, C_in, aux, first element, element count}
//   kind 0 / 1: paired bf16 image of [C_out][C_in] / of a transposed source;  2 / 3: the same as fp16
//   kind 4: depthwise taps [C][K3] -> [K3][C] fp32 (aux = K3, C_out = C);      5: the same with the stencil reversed
//   kind 6 / 7: plain row-major bf16 [C_out][C_in] of the matrix / of a transposed source (pytc_pw_conv_fwd with w_paired = 2)
__global__ void __launch_bounds__(256)
pack_multi_kernel(const long* __restrict__ table, int n_items, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // the item of the block's first element, found once per workgroup (every thread walking the table by itself was eight dependent
  // loads per output element: 86 us per MedNeXt-S step); a thread then steps forward over the few item boundaries inside the block
  __shared__ int s_lo;
  if (threadIdx.x == 0) {
    const long i0 = (long)blockIdx.x * blockDim.x;
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {                                 // last item whose first element is <= i0
      const int mid = (lo + hi + 1) >> 1;
      if (table[mid * 8 + 6] <= i0) lo = mid; else hi = mid - 1;
    }
    s_lo = lo;
  }
  __syncthreads();
  if (i >= total) return;
  int lo = s_lo;
  while (lo + 1 < n_items && table[(lo + 1) * 8 + 6] <= i) ++lo;
  const long* it = table + lo * 8;
  const float* w = reinterpret_cast<const float*>(it[0]);
  const int kind = (int)it[2], C_out = (int)it[3], C_in = (int)it[4], aux = (int)it[5];
  const long e = i - it[6];
  if (kind >= 6) {      // 6 / 7: plain row-major bf16 [C_out][C_in] of the matrix / of a transposed source (the LDS-tiled GEMM's weights)
    const int o = (int)(e / C_in), k = (int)(e % C_in);
    reinterpret_cast<bf16_t*>(it[1])[e] = from_f32<bf16_t>(kind == 7 ? w[(long)k * C_out + o] : w[e]);
    return;
  }
  if (kind >= 4) {
    const int K3 = aux, C = C_out;
    const int k = (int)(e / C), c = (int)(e % C);
    reinterpret_cast<float*>(it[1])[e] = w[(long)c * K3 + (kind == 5 ? K3 - 1 - k : k)];
    return;
  }
  const int KG = aux, transposed = kind & 1;
  const int j = (int)(e % 8);
  long t = e / 8;
  const int lane = (int)(t % 64);
  t /= 64;
  const int kg = (int)(t % KG);
  const int T = (int)(t / KG);
  const int o = paired_row(T, lane & 15);
  const int k = kg * 32 + (lane >> 4) * 8 + j;
  float v = 0.f;
  if (o < C_out && k < C_in) v = transposed ? w[(long)k * C_out + o] : w[(long)o * C_in + k];
  if (kind >= 2) reinterpret_cast<_Float16*>(it[1])[e] = (_Float16)v;
  else reinterpret_cast<bf16_t*>(it[1])[e] = from_f32<bf16_t>(v);
}

// GroupNorm finalize + fold into the expanding conv of the mixer that follows (see MlpParams::w2_stride): one workgroup of
// 1024 threads per sample -- C channel lanes x 1024/C slot lanes reduce the (sum, sum of squares) slots in a fixed order, the
// affine goes to LDS, and every thread then writes its share of the sample's paired bf16 image of W2 * diag(a) and of the
// folded bias.  Takes the place of groupnorm_finalize_kernel (same latency class: the slot loop dominates).
__global__ void __launch_bounds__(1024)
groupnorm_fold_mlp_kernel(const float* __restrict__ stats, int slots, float count, const float* __restrict__ gamma,
                          const float* __restrict__ beta, float eps, const float* __restrict__ w2, const float* __restrict__ b2,
                          bf16_t* __restrict__ w2n, float* __restrict__ b2n, float* __restrict__ ab_out, int C, int C_hid) {
  // slot reduction: a thread owns 4 consecutive channels of one of (sum | sum of squares) and every (1024 / (C/2))-th slot -- 16-byte
  // loads, 64 (C = 32) ... 16 (C = 128) slot lanes, so the dependent-load chain of the level-0 launch is 1176 / 64 = 19 steps (the
  // first cut: 4-byte loads, 32 lanes, 37 steps of two loads each = 12.7 us against groupnorm_finalize's 4.9); the lane partials meet
  // in LDS in two fixed-order stages
  __shared__ __attribute__((aligned(16))) float red[1024 * 4];
  __shared__ float part[8][2 * 128];
  __shared__ float sa[128], sb[128], smean[128], sbeta[128];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int Q = C / 2;                        // float4 columns of one slot row: [sum C | sum of squares C]
  const int col = tid % Q, sl = tid / Q, SL = 1024 / Q;
  const float* base = stats + (long)n * slots * 2 * C + col * 4;
  // the weight pieces this thread will scale do not depend on the statistics: request them first (their L2 latency rides under the
  // slot loads), then ALL slot rows of a pass before the first addition -- one memory round trip per 20 * SL slots instead of a
  // dependent chain (additions in slot order as before: same bits)
  const int KG = C / 32;
  const long total = (long)C_hid * C;
  constexpr int PRE = 2;
  f32x4_t wpre[PRE][2];
#pragma unroll
  for (int i = 0; i < PRE; ++i) {
    const long e8 = tid + (long)i * 1024;
    if (e8 < total / 8) {
      long t = e8;
      const int lane = (int)(t % 64);
      t /= 64;
      const int kg = (int)(t % KG);
      const int o = paired_row((int)(t / KG), lane & 15);
      const float* src = w2 + (long)o * C + kg * 32 + (lane >> 4) * 8;
      wpre[i][0] = *reinterpret_cast<const f32x4_t*>(src);
      wpre[i][1] = *reinterpret_cast<const f32x4_t*>(src + 4);
    }
  }
  f32x4_t a = {0.f, 0.f, 0.f, 0.f};
  constexpr int INFL = 20;
  for (int s0 = sl; s0 < slots; s0 += SL * INFL) {
    f32x4_t v[INFL];
#pragma unroll
    for (int i = 0; i < INFL; ++i) {
      const int s = s0 + i * SL;
      v[i] = s < slots ? *reinterpret_cast<const f32x4_t*>(base + (long)s * 2 * C) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < INFL; ++i)
      if (s0 + i * SL < slots) a += v[i];
  }
  *reinterpret_cast<f32x4_t*>(&red[tid * 4]) = a;             // red[sl][2C] as float4 columns
  __syncthreads();
  {
    // stage 1: 8 partial sums per (which, channel) over SL / 8 lanes each; stage 2 below adds the 8 in order
    const int e = tid % (2 * C), pg = tid / (2 * C);           // 2C <= 256 elements, up to 1024 / (2C) >= 4 part groups
    const int groups = 1024 / (2 * C) < 8 ? 1024 / (2 * C) : 8;
    if (pg < groups) {
      float t = 0.f;
      for (int s = pg; s < SL; s += groups) t += red[s * 2 * C + e];
      part[pg][e] = t;
    }
    __syncthreads();
    if (tid < C) {
      float t1 = 0.f, t2 = 0.f;
      for (int g = 0; g < groups; ++g) { t1 += part[g][tid]; t2 += part[g][C + tid]; }
      const float mean = t1 / count;
      const float var = fmaxf(t2 / count - mean * mean, 0.f);
      float rstd = rsqrtf(var + eps);
      rstd = rstd * (1.5f - 0.5f * (var + eps) * rstd * rstd);      // one Newton step: rsqrtf is approximate
      const float av = (gamma ? gamma[tid] : 1.f) * rstd;
      const float bv = (beta ? beta[tid] : 0.f) - mean * av;
      sa[tid] = av;
      sb[tid] = bv;
      smean[tid] = mean;
      sbeta[tid] = beta ? beta[tid] : 0.f;
      if (ab_out) {
        ab_out[((long)n * 2 + 0) * C + tid] = av;
        ab_out[((long)n * 2 + 1) * C + tid] = bv;
      }
    }
  }
  __syncthreads();
  bf16_t* img = w2n + (long)n * total;
  for (long e8 = tid, i = 0; e8 < total / 8; e8 += 1024, ++i) {   // one 16-byte fragment piece (8 consecutive k of one output row) per step
    long t = e8;
    const int lane = (int)(t % 64);
    t /= 64;
    const int kg = (int)(t % KG);
    const int T = (int)(t / KG);
    const int o = paired_row(T, lane & 15);
    const int k = kg * 32 + (lane >> 4) * 8;
    float v[8];
    if (i < PRE) {
#pragma unroll
      for (int q = 0; q < PRE; ++q)
        if (q == i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = wpre[q][0][j]; v[4 + j] = wpre[q][1][j]; }
        }
    } else {
      VecIO<float, 4>::load(w2 + (long)o * C + k, reinterpret_cast<float(&)[4]>(v[0]));
      VecIO<float, 4>::load(w2 + (long)o * C + k + 4, reinterpret_cast<float(&)[4]>(v[4]));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= sa[k + j];
    *reinterpret_cast<bf16x8_t*>(img + e8 * 8) = Mma<bf16_t>::from_floats(v);
  }
  // Folded bias, centred with the weights the mixer will actually multiply by:  W2 (a (t - mean) + beta) + b2 = sum_k bf16(W2 a)_k (t_k - mean_k)
  // + sum_k W2_k beta_k + b2 up to the rounding of (W2 a)_k TIMES (t_k - mean_k) -- the spread of the channel, not its offset.  (The first
  // cut used b2 + W2 (beta - mean a) with the UNROUNDED product: the rounding error of every folded weight then met the channel's MEAN, an
  // error that grows like |mean| / std relative to the affine-prologue form -- ADVICE r04.)  k ascending, fixed order.
  for (int o = tid; o < C_hid; o += 1024) {
    float acc = b2 ? b2[o] : 0.f;
    const float* wr = w2 + (long)o * C;
    for (int k = 0; k < C; k += 8) {                  // 16-byte loads (a lane walks its own row: 64 lines per load instruction), k ascending
      const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wr + k), w1 = *reinterpret_cast<const f32x4_t*>(wr + k + 4);
      float v[8], wn[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = w0[j] * sa[k + j]; v[4 + j] = w1[j] * sa[k + 4 + j]; }
      const bf16x8_t rounded = Mma<bf16_t>::from_floats(v);           // the very conversion that produced the image above
      VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rounded), wn);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float wj = j < 4 ? w0[j] : w1[j - 4];
        acc = fmaf(wj, sbeta[k + j], acc);
        acc = fmaf(-wn[j], smean[k + j], acc);
      }
    }
    b2n[(long)n * C_hid + o] = acc;
  }
}

// one-time upload of the GELU table (host double precision; blocking copy, first mixer launch of the process)
static bool ensure_gelu_lut() {
  static int state = 0;     // 0 = not tried, 1 = ready, -1 = failed
  if (state == 0) {
    static float2 host[GELU_LUT_N];
    auto f = [](double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440)); };
    for (int i = 0; i < GELU_LUT_N; ++i) {
      const double x0 = -8.0 + i / 64.0, x1 = x0 + 1.0 / 64.0;
      const double b = (f(x1) - f(x0)) * 64.0;
      host[i].x = (float)(f(x0) - b * x0);
      host[i].y = (float)b;
    }
    state = hipMemcpyToSymbol(HIP_SYMBOL(g_gelu_lut), host, sizeof(host)) == hipSuccess ? 1 : -1;
  }
  return state == 1;
}

template <int KS_IN, int MO, int NT>
static void launch_mlp(const MlpParams& p, int N, hipStream_t s) {
  long rows_per_block = 4L * NT * 16;
  dim3 grid((unsigned)((p.rps + rows_per_block - 1) / rows_per_block), (unsigned)N), block(256);
  if (p.hp_in) {    // training backward: both data-gradient GEMMs with GELU' between them
    hipLaunchKernelGGL((pw_mlp_kernel<KS_IN, MO, NT, 1, false, false, false, true>), grid, block, 0, s, p);
    return;
  }
  if (p.hp || p.round_h) {       // training forward: fast GELU (what the backward kernels differentiate), hidden pre-activation stored / rounded
    if (p.w3_f16) hipLaunchKernelGGL((pw_mlp_kernel<KS_IN, MO, NT, 3, false, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((pw_mlp_kernel<KS_IN, MO, NT, 1, false, false, true>), grid, block, 0, s, p);
    return;
  }
  if (p.w3_f16) {
    hipLaunchKernelGGL((pw_mlp_kernel<KS_IN, MO, NT, 3>), grid, block, 0, s, p);
    return;
  }
  if (tuning_get("mlp_exact_gelu", 0))
    hipLaunchKernelGGL((pw_mlp_kernel<KS_IN, MO, NT, 0>), grid, block, 0, s, p);
  else if (tuning_get("mlp_gelu_lut", 0) && ensure_gelu_lut())
    hipLaunchKernelGGL((pw_mlp_kernel<KS_IN, MO, NT, 2>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((pw_mlp_kernel<KS_IN, MO, NT, 1>), grid, block, 0, s, p);
}

// (C_in/32, C_out/16) pairs that occur in MedNeXt with 32 base channels: same-res, down (x2), up (/2)
static bool dispatch_mlp(const MlpParams& p, int N, hipStream_t s) {
  const int ks = p.C_in / 32, mo = p.C_out / 16;
  const int variant = tuning_get("mlp_variant", 0);
#define PYTC_MLP_CASE(KS, MOO, NTT) \
  if (ks == KS && mo == MOO) { launch_mlp<KS, MOO, NTT>(p, N, s); return true; }
  // voxel tiles per wave (NT) chosen per shape from tools/kbench.py measurements on MI355X
  if (!(variant & 1)) { PYTC_MLP_CASE(2, 4, 2) }
  if (variant & 2) { PYTC_MLP_CASE(1, 2, 2) PYTC_MLP_CASE(2, 2, 2) }
  if (variant & 4) { PYTC_MLP_CASE(1, 4, 2) }
  PYTC_MLP_CASE(1, 2, 4)
  PYTC_MLP_CASE(1, 4, 4)
  PYTC_MLP_CASE(2, 2, 4)
  PYTC_MLP_CASE(2, 4, 4)
  PYTC_MLP_CASE(2, 8, 2)
  PYTC_MLP_CASE(4, 4, 2)
  PYTC_MLP_CASE(4, 8, 2)
  PYTC_MLP_CASE(4, 16, 1)
  PYTC_MLP_CASE(8, 8, 2)
  PYTC_MLP_CASE(8, 16, 1)
  PYTC_MLP_CASE(8, 32, 1)
  PYTC_MLP_CASE(16, 16, 1)
  PYTC_MLP_CASE(16, 32, 1)
#undef PYTC_MLP_CASE
  return false;
}

static bool mlp_shape_ok(int C_in, int C_hid, int C_out) {
  if (C_in % 32 || C_hid % 32 || C_out % 32 || C_in < 32 || C_out < 32 || C_hid < 32) return false;
  const int ks = C_in / 32, mo = C_out / 16;
  static const int ok[][2] = {{1, 2}, {1, 4}, {2, 2}, {2, 4}, {2, 8}, {4, 4}, {4, 8}, {4, 16}, {8, 8},
                              {8, 16}, {8, 32}, {16, 16}, {16, 32}};
  for (auto& c : ok)
    if (c[0] == ks && c[1] == mo) return true;
  return false;
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_pw_mlp_supported(int C_in, int C_hid, int C_out) { return mlp_shape_ok(C_in, C_hid, C_out) ? 1 : 0; }

extern "C" int pytc_groupnorm_fold_mlp(const float* stats, int slots, float count, const float* gamma, const float* beta,
                                       float eps, const float* w2, const float* b2, void* w2n, float* b2n, float* ab_out, int N,
                                       int C, int C_hid, void* stream) {
  PYTC_REQUIRE(stats && w2 && w2n && b2n && slots >= 1 && count > 0 && N >= 1, "groupnorm_fold_mlp: bad arguments");
  PYTC_REQUIRE(C == 32 || C == 64 || C == 128, "groupnorm_fold_mlp: C=%d (32, 64 or 128)", C);
  PYTC_REQUIRE(C_hid % 32 == 0 && C_hid >= 32 && C_hid <= 512, "groupnorm_fold_mlp: C_hid=%d", C_hid);
  hipLaunchKernelGGL(groupnorm_fold_mlp_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, stats, slots, count, gamma, beta,
                     eps, w2, b2, (bf16_t*)w2n, b2n, ab_out, C, C_hid);
  PYTC_LAUNCH_CHECK("groupnorm_fold_mlp");
  return PYTC_OK;
}

extern "C" int pytc_pw_mlp_head_supported(int C_in, int C_hid, int C_out) {
  return (C_out == 32 && (C_in == 32 || C_in == 64) && C_hid % 32 == 0 && C_hid >= 32) ? 1 : 0;
}

extern "C" int pytc_pw_pack_weight_paired(const float* w, int C_out, int C_in, int transposed, void* packed,
                                          void* stream) {
  PYTC_REQUIRE(w && packed && C_out >= 1 && C_in >= 1, "pw_pack_weight_paired: bad arguments");
  PYTC_REQUIRE(C_out % 32 == 0, "pw_pack_weight_paired: C_out=%d must be a multiple of 32", C_out);
  long total = pytc_pw_packed_elems(C_out, C_in, PYTC_BF16);
  int KG = (C_in + 31) / 32;
  hipLaunchKernelGGL(pw_pack_paired_kernel<bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, C_out,
                     C_in, transposed, (bf16_t*)packed, KG, total);
  PYTC_LAUNCH_CHECK("pw_pack_weight_paired");
  return PYTC_OK;
}

extern "C" int pytc_pw_pack_weight_paired_f16(const float* w, int C_out, int C_in, int transposed, void* packed,
                                              void* stream) {
  PYTC_REQUIRE(w && packed && C_out >= 1 && C_in >= 1, "pw_pack_weight_paired_f16: bad arguments");
  PYTC_REQUIRE(C_out % 32 == 0, "pw_pack_weight_paired_f16: C_out=%d must be a multiple of 32", C_out);
  long total = pytc_pw_packed_elems(C_out, C_in, PYTC_BF16);       // same element count, 2 bytes each
  int KG = (C_in + 31) / 32;
  hipLaunchKernelGGL(pw_pack_paired_kernel<_Float16>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     C_out, C_in, transposed, (_Float16*)packed, KG, total);
  PYTC_LAUNCH_CHECK("pw_pack_weight_paired_f16");
  return PYTC_OK;
}

extern "C" int pytc_pack_multi(const int64_t* table_dev, int n_items, int64_t total_elems, void* stream) {
  PYTC_REQUIRE(table_dev && n_items >= 1 && total_elems >= 1, "pack_multi: bad arguments");
  hipLaunchKernelGGL(pack_multi_kernel, dim3(ceil_div(total_elems, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long*>(table_dev), n_items, (long)total_elems);
  PYTC_LAUNCH_CHECK("pack_multi");
  return PYTC_OK;
}

extern "C" int pytc_pw_mlp_head_fwd(const pytc_mlp_args* a, const void* head_w, const float* head_b, float* head_y,
                                    int n_head, int store_y, void* stream) {
  PYTC_REQUIRE(a && a->t && (a->ab || a->per_sample) && a->w2_packed && a->w3_packed && a->b2 && a->b3 && head_w && head_y, "pw_mlp_head: null pointer");
  PYTC_REQUIRE(!(a->ab && a->per_sample), "pw_mlp_head: per-sample (norm-folded) expand operands come without an affine");
  PYTC_REQUIRE(!store_y || a->y, "pw_mlp_head: store_y without an output buffer");
  PYTC_REQUIRE(a->N >= 1 && a->rows_per_sample >= 1 && n_head >= 1 && n_head <= 16, "pw_mlp_head: bad shape");
  if (!pytc_pw_mlp_head_supported(a->C_in, a->C_hid, a->C_out)) {
    set_error("pw_mlp_head: no fused kernel for C_in=%d C_hid=%d C_out=%d", a->C_in, a->C_hid, a->C_out);
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || (a->res_mode == PYTC_RES_ADD && a->res), "pw_mlp_head: residual add or none");
  MlpParams p;
  p.t = (const bf16_t*)a->t; p.ab = a->ab; p.w2 = (const bf16x8_t*)a->w2_packed; p.b2 = a->b2;
  p.w3 = (const bf16x8_t*)a->w3_packed; p.b3 = a->b3;
  p.rps = a->rows_per_sample; p.C_in = a->C_in; p.C_hid = a->C_hid; p.C_out = a->C_out; p.HC = a->C_hid / 32;
  p.w2_stride = (long)(a->C_hid / 16) * (a->C_in / 32) * 64;
  p.e.res = a->res; p.e.res_low = nullptr; p.e.res_bias = nullptr; p.e.y = a->y;
  p.e.rps_out = a->rows_per_sample; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode;
  p.e.nt = stream_nt_policy((long)a->N * a->rows_per_sample * (a->C_in > a->C_out ? a->C_in : a->C_out) * 2);
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  p.head_w = (const bf16x8_t*)head_w; p.head_b = head_b; p.head_y = head_y; p.n_head = n_head; p.store_y = store_y;
  p.w3_f16 = a->w3_format == PYTC_W3_F16 ? 1 : 0;
  dim3 grid((unsigned)((p.rps + 4L * 4 * 16 - 1) / (4L * 4 * 16)), (unsigned)a->N), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool exact = tuning_get("mlp_exact_gelu", 0) != 0;
  if (mlp_dma_applies(a, p)) {
    mlp_dma_launch<2>(a, p, s);
    PYTC_LAUNCH_CHECK("pw_mlp_head_dma");
    return PYTC_OK;
  }
  if (p.w3_f16) {
    if (a->C_in == 32) hipLaunchKernelGGL((pw_mlp_kernel<1, 2, 4, 3, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((pw_mlp_kernel<2, 2, 4, 3, true>), grid, block, 0, s, p);
  } else if (a->C_in == 32) {
    if (exact) hipLaunchKernelGGL((pw_mlp_kernel<1, 2, 4, 0, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((pw_mlp_kernel<1, 2, 4, 1, true>), grid, block, 0, s, p);
  } else {
    if (exact) hipLaunchKernelGGL((pw_mlp_kernel<2, 2, 4, 0, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((pw_mlp_kernel<2, 2, 4, 1, true>), grid, block, 0, s, p);
  }
  PYTC_LAUNCH_CHECK("pw_mlp_head");
  return PYTC_OK;
}

extern "C" int pytc_pw_mlp_stemres_fwd(const pytc_mlp_args* a, const float* stem_x, const float* stem_w, const float* stem_b,
                                       void* stream) {
  PYTC_REQUIRE(a && a->t && (a->ab || a->per_sample) && a->w2_packed && a->w3_packed && a->b2 && a->b3 && a->y && stem_x && stem_w && stem_b,
               "pw_mlp_stemres: null pointer");
  PYTC_REQUIRE(!(a->ab && a->per_sample), "pw_mlp_stemres: per-sample (norm-folded) expand operands come without an affine");
  PYTC_REQUIRE(a->N >= 1 && a->rows_per_sample >= 1, "pw_mlp_stemres: bad shape");
  if (!(a->C_in == 32 && a->C_out == 32 && a->C_hid % 32 == 0 && a->C_hid >= 32)) {
    set_error("pw_mlp_stemres: no fused kernel for C_in=%d C_hid=%d C_out=%d", a->C_in, a->C_hid, a->C_out);
    return PYTC_ERR_UNSUPPORTED;
  }
  MlpParams p{};
  p.t = (const bf16_t*)a->t; p.ab = a->ab; p.w2 = (const bf16x8_t*)a->w2_packed; p.b2 = a->b2;
  p.w3 = (const bf16x8_t*)a->w3_packed; p.b3 = a->b3;
  p.rps = a->rows_per_sample; p.C_in = a->C_in; p.C_hid = a->C_hid; p.C_out = a->C_out; p.HC = a->C_hid / 32;
  p.w2_stride = (long)(a->C_hid / 16) * (a->C_in / 32) * 64;
  p.e.res = a->y;              // never dereferenced (the residual rows are recomputed); non-null for the epilogue's checks
  p.e.y = a->y;
  p.e.rps_out = a->rows_per_sample; p.e.C_out = a->C_out; p.e.res_mode = PYTC_RES_ADD;
  p.e.nt = stream_nt_policy((long)a->N * a->rows_per_sample * (a->C_in > a->C_out ? a->C_in : a->C_out) * 2);
  p.stem_x = stem_x; p.stem_w = stem_w; p.stem_b = stem_b;
  p.w3_f16 = a->w3_format == PYTC_W3_F16 ? 1 : 0;
  dim3 grid((unsigned)((p.rps + 4L * 4 * 16 - 1) / (4L * 4 * 16)), (unsigned)a->N), block(256);
  hipStream_t s = (hipStream_t)stream;
  pytc_mlp_args shape = *a;
  shape.res_mode = PYTC_RES_NONE;          // this entry ignores a->res_mode (the residual is the recomputed stem row)
  if (mlp_dma_applies(&shape, p)) {
    mlp_dma_launch<1>(a, p, s);
    PYTC_LAUNCH_CHECK("pw_mlp_stemres_dma");
    return PYTC_OK;
  }
  if (p.w3_f16) hipLaunchKernelGGL((pw_mlp_kernel<1, 2, 4, 3, false, true>), grid, block, 0, s, p);
  else if (tuning_get("mlp_exact_gelu", 0) != 0) hipLaunchKernelGGL((pw_mlp_kernel<1, 2, 4, 0, false, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((pw_mlp_kernel<1, 2, 4, 1, false, true>), grid, block, 0, s, p);
  PYTC_LAUNCH_CHECK("pw_mlp_stemres");
  return PYTC_OK;
}

// Which kernel pytc_pw_mlp_fwd / _head_fwd / _stemres_fwd will launch for these arguments: 1 = the DMA-prefetching form (pw_mlp_dma_kernel), 0 = the
// one-tile-per-wave kernel.  For profilers and bench tables that name launches by device symbol; `ignore_res_mode` = the stem-residual entry.
extern "C" int pytc_pw_mlp_dma_applies(const pytc_mlp_args* a, int ignore_res_mode) {
  if (!a) return 0;
  MlpParams p{};
  p.HC = a->C_hid / 32;
  p.w3_f16 = a->w3_format == PYTC_W3_F16 ? 1 : 0;
  pytc_mlp_args shape = *a;
  if (ignore_res_mode) shape.res_mode = PYTC_RES_NONE;
  return mlp_dma_applies(&shape, p) ? 1 : 0;
}

// The mixer of a 32-channel block with a 32 -> 32 1x1x1 conv of its (bf16-rounded) output in the epilogue: z = bf16(W y + b).  proj_w: the conv's
// paired bf16 image (pytc_pw_pack_weight_paired, C_out = C_in = 32); y itself is written only when store_y.  Folded operands, fp16 projection
// image of the mixer, hidden width 64 / 96 / 128 (the shapes of pw_mlp_dma_kernel; any row count).
extern "C" int pytc_pw_mlp_proj_supported(int C_in, int C_hid, int C_out, int C_proj) {
  return (C_in == 32 && C_out == 32 && C_proj == 32 && (C_hid == 64 || C_hid == 96 || C_hid == 128)) ? 1 : 0;
}
extern "C" int pytc_pw_mlp_proj_fwd(const pytc_mlp_args* a, const void* proj_w, const float* proj_b, void* z, int store_y, void* stream) {
  PYTC_REQUIRE(a && a->t && a->w2_packed && a->w3_packed && a->b2 && a->b3 && proj_w && z, "pw_mlp_proj: null pointer");
  PYTC_REQUIRE(a->per_sample && !a->ab, "pw_mlp_proj: takes the per-sample (norm-folded) expand operands");
  PYTC_REQUIRE(a->w3_format == PYTC_W3_F16, "pw_mlp_proj: the mixer's projection image must be fp16 (pytc_pw_pack_weight_paired_f16)");
  PYTC_REQUIRE(!store_y || a->y, "pw_mlp_proj: store_y without an output buffer");
  PYTC_REQUIRE(a->N >= 1 && a->rows_per_sample >= 1, "pw_mlp_proj: bad shape");
  if (!pytc_pw_mlp_proj_supported(a->C_in, a->C_hid, a->C_out, 32)) {
    set_error("pw_mlp_proj: no fused kernel for C_in=%d C_hid=%d C_out=%d", a->C_in, a->C_hid, a->C_out);
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || (a->res_mode == PYTC_RES_ADD && a->res), "pw_mlp_proj: residual add or none");
  MlpParams p{};
  p.t = (const bf16_t*)a->t; p.ab = nullptr; p.w2 = (const bf16x8_t*)a->w2_packed; p.b2 = a->b2;
  p.w3 = (const bf16x8_t*)a->w3_packed; p.b3 = a->b3;
  p.rps = a->rows_per_sample; p.C_in = a->C_in; p.C_hid = a->C_hid; p.C_out = a->C_out; p.HC = a->C_hid / 32;
  p.w2_stride = (long)(a->C_hid / 16) * (a->C_in / 32) * 64;
  p.e.res = a->res; p.e.y = a->y; p.e.rps_out = a->rows_per_sample; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode;
  p.head_w = (const bf16x8_t*)proj_w; p.head_b = proj_b; p.proj_y = (bf16_t*)z; p.store_y = store_y;
  p.w3_f16 = 1;
  mlp_dma_launch<3>(a, p, (hipStream_t)stream);
  PYTC_LAUNCH_CHECK("pw_mlp_proj");
  return PYTC_OK;
}

static int mlp_fwd_impl(const pytc_mlp_args* a, void* hp, void* stream, const void* hp_in = nullptr, bool round_h = false);

// a->t = dY [N][rows][C_in], a->w2_packed = W3^T image (C_in -> C_hid), a->w3_packed = W2^T image (C_hid -> C_out),
// a->ab = identity affine, a->b2 / a->b3 = zeros, a->y = dX; hidden_pre = the forward's stored pre-activation,
// d_hidden receives (W3^T dY) * GELU'(hidden_pre)
extern "C" int pytc_pw_mlp_bwd(const pytc_mlp_args* a, const void* hidden_pre, void* d_hidden, void* stream) {
  PYTC_REQUIRE(hidden_pre && d_hidden, "pw_mlp_bwd: null hidden buffers");
  PYTC_REQUIRE(a && a->res_mode == PYTC_RES_NONE, "pw_mlp_bwd: no residual in the backward mixer");
  return mlp_fwd_impl(a, d_hidden, stream, hidden_pre);
}

extern "C" int pytc_pw_mlp_train_fwd(const pytc_mlp_args* a, void* hidden_pre, void* stream) {
  PYTC_REQUIRE(hidden_pre, "pw_mlp_train: null hidden buffer");
  return mlp_fwd_impl(a, hidden_pre, stream);
}

// the training forward without the store: the hidden pre-activation is rounded to bf16 before the activation exactly as
// pytc_pw_mlp_train_fwd does, y carries the same bits, nothing is written besides y (pytc_mixer_bwd_rc rebuilds the hidden tensor)
extern "C" int pytc_pw_mlp_train_fwd_nostore(const pytc_mlp_args* a, void* stream) {
  return mlp_fwd_impl(a, nullptr, stream, nullptr, true);
}

extern "C" int pytc_pw_mlp_fwd(const pytc_mlp_args* a, void* stream) { return mlp_fwd_impl(a, nullptr, stream); }

static int mlp_fwd_impl(const pytc_mlp_args* a, void* hp, void* stream, const void* hp_in, bool round_h) {
  PYTC_REQUIRE(a && a->t && (a->ab || a->per_sample) && a->w2_packed && a->w3_packed && a->b2 && a->b3 && a->y, "pw_mlp: null pointer");
  PYTC_REQUIRE(!(a->ab && a->per_sample), "pw_mlp: per-sample (norm-folded) expand operands come without an affine");
  PYTC_REQUIRE(!(a->per_sample && (hp || hp_in || round_h)), "pw_mlp: the training kernels take the shared expand image and the affine");
  PYTC_REQUIRE(a->N >= 1 && a->rows_per_sample >= 1, "pw_mlp: bad shape");
  if (!mlp_shape_ok(a->C_in, a->C_hid, a->C_out)) {
    set_error("pw_mlp: no fused kernel for C_in=%d C_hid=%d C_out=%d", a->C_in, a->C_hid, a->C_out);
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || a->res, "pw_mlp: residual mode without residual pointer");
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || a->res_mode == PYTC_RES_ADD || a->res_mode == PYTC_RES_UPSAMPLE,
               "pw_mlp: unsupported res_mode %d", a->res_mode);
  MlpParams p{};
  p.t = (const bf16_t*)a->t; p.ab = a->ab; p.w2 = (const bf16x8_t*)a->w2_packed; p.b2 = a->b2;
  p.w3 = (const bf16x8_t*)a->w3_packed; p.b3 = a->b3;
  p.rps = a->rows_per_sample; p.C_in = a->C_in; p.C_hid = a->C_hid; p.C_out = a->C_out; p.HC = a->C_hid / 32;
  p.w2_stride = (long)(a->C_hid / 16) * (a->C_in / 32) * 64;
  p.e.res = a->res; p.e.res_low = a->res_low; p.e.res_bias = a->res_bias; p.e.y = a->y;
  p.e.rps_out = a->rows_per_sample; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode;
  p.e.nt = stream_nt_policy((long)a->N * a->rows_per_sample * (a->C_in > a->C_out ? a->C_in : a->C_out) * 2);
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  p.hp = (bf16_t*)hp;
  p.hp_in = (const bf16_t*)hp_in;
  p.round_h = round_h ? 1 : 0;
  p.w3_f16 = a->w3_format == PYTC_W3_F16 ? 1 : 0;
  PYTC_REQUIRE(!(p.w3_f16 && hp_in), "pw_mlp_bwd: the backward mixer takes bf16 weight images");
  if (a->res_mode == PYTC_RES_UPSAMPLE) {
    PYTC_REQUIRE((long)a->Di * a->Hi * a->Wi == a->rows_per_sample && !(a->Di & 1) && !(a->Hi & 1) && !(a->Wi & 1),
                 "pw_mlp: RES_UPSAMPLE needs the (even) output grid");
    PYTC_REQUIRE(a->rows_per_sample < (1L << 31), "pw_mlp: RES_UPSAMPLE positions are 32-bit (rows per sample < 2^31)");
    p.e.Go_d = a->Di; p.e.Go_h = a->Hi; p.e.Go_w = a->Wi;
    p.e.Gl_d = a->Di / 2; p.e.Gl_h = a->Hi / 2; p.e.Gl_w = a->Wi / 2;
  }
  // level-0 shapes of the inference mixers with per-sample operands: the DMA-prefetching form (pw_mlp_dma_kernel)
  if (!hp && !hp_in && !round_h && mlp_dma_applies(a, p)) {
    mlp_dma_launch<0>(a, p, (hipStream_t)stream);
    PYTC_LAUNCH_CHECK("pw_mlp_dma");
    return PYTC_OK;
  }
  if (!dispatch_mlp(p, a->N, (hipStream_t)stream)) {
    set_error("pw_mlp: dispatch failed");
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_LAUNCH_CHECK("pw_mlp");
  return PYTC_OK;
}
