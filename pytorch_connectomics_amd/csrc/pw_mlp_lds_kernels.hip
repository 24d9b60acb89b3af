// Fused MedNeXt channel mixer with the WEIGHTS RESIDENT IN LDS (round 4) -- the mid levels of the network.
//
// pw_mlp_kernel streams every weight fragment of every hidden chunk from L2, per wave: at level 0 (32 / 64 hidden channels) that
// is nothing, at the deep levels the GEMM pair of pw_gemm_kernels.hip took over, but in between -- 64->128->64 and 128->256->64 at
// 56^3, 128->256->128 at 28^3 -- a wave of 32 rows pulls 32 ... 130 KB of fragments through dependent loads (128->256->64 on 8 x 56^3:
// 4.3 GB of L2 traffic per launch, 404 us against a 140 us byte floor and 71 us of GELU issue).  Here a workgroup copies both paired
// weight images into LDS ONCE (32 ... 128 KB) and then walks its share of the row tiles, sample by sample: fragments are
// ds_read_b128 at lane-contiguous addresses (conflict-free), no barrier inside the row loop, activations exactly as before.
// The arithmetic is pw_mlp_kernel<..., GELU_MODE = 3>'s, instruction for instruction (same MFMA order, same packed-fp16 GELU, same
// epilogue): results are bit-identical (tests/test_gpu_kernels.py::test_lds_resident_mixer_is_bit_identical).  Per-sample folded
// expand operands (pytc_groupnorm_fold_mlp) are supported: the W2 image is re-staged when the sample changes.
#include <mutex>

#include "pw_common.h"

namespace pytc {

struct MlpLdsParams {
  const bf16_t* t;
  const float* ab;            // [N][2][C_in] or null (folded operands)
  const bf16x8_t* w2;         // paired bf16 image(s) [hid/16][C_in/32][64 lanes][8]; N of them when folded
  const float* b2;            // [C_hid] ([N][C_hid] when folded)
  const h8_t* w3;             // paired fp16 image [C_out/16][hid/32][64][8]
  const float* b3;
  EpiParams e;
  long rps;
  int N, C_in, C_hid, C_out, HC;
  long w2_stride;             // bf16x8 elements per sample image (folded)
};

// One wave's operand tile: the raw input rows (16 bytes per lane per k-step) and the residual / skip rows.
template <int KS_IN, int MO, int NT>
struct LdsTile {
  uint4 raw[KS_IN][NT];
  uint4 res[MO / 2][NT];
};

template <int KS_IN, int MO, int NT>
__device__ __forceinline__ void lds_tile_load(LdsTile<KS_IN, MO, NT>& T, const MlpLdsParams& p, int n, long row0, int r, int kb,
                                              bool with_res) {
  const bf16_t* tn = p.t + (long)n * p.rps * (KS_IN * 32);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const long o = row0 + nt * 16 + r;
    const long rr = o < p.rps ? o : p.rps - 1;
#pragma unroll
    for (int ks = 0; ks < KS_IN; ++ks) T.raw[ks][nt] = ld_stream(reinterpret_cast<const uint4*>(tn + rr * (KS_IN * 32) + ks * 32 + kb * 8), p.e.nt);
  }
  if (with_res) {
    const bf16_t* resn = reinterpret_cast<const bf16_t*>(p.e.res) + (long)n * p.rps * (MO * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const long o = row0 + nt * 16 + r;
      const long rr = o < p.rps ? o : p.rps - 1;
#pragma unroll
      for (int pr = 0; pr < MO / 2; ++pr) T.res[pr][nt] = ld_stream(reinterpret_cast<const uint4*>(resn + rr * (MO * 16) + pr * 32 + kb * 8), p.e.nt);
    }
  }
}

// NWAVES waves per workgroup, compiled for WPS waves per SIMD (WPS * 4 / NWAVES workgroups per CU).
// Measured and removed (profiles/r04_lds_resident_mixer.txt): a wave requesting the rows of its NEXT tile before it computes the
// current one (software prefetch; the registers are there at 2 waves per SIMD) -- never faster than the plain loop at the same
// occupancy (128->256->64: 282 vs 276 us) and much slower at 3 waves per SIMD (434 vs 244 us, spills).
template <int KS_IN, int MO, int NT, int NWAVES, int WPS>
__global__ void __launch_bounds__(NWAVES * 64, WPS)
pw_mlp_lds_kernel(MlpLdsParams p) {
  constexpr int CIN = KS_IN * 32, COUT = MO * 16;       // compile-time row pitches: shifts, not 64-bit multiplies
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bf16x8_t* lw2 = reinterpret_cast<bf16x8_t*>(lds_raw);                                   // [HC*2][KS_IN][64]
  h8_t* lw3 = reinterpret_cast<h8_t*>(lds_raw + (size_t)p.C_hid * CIN * 2);             // [MO][HC][64]
  float* lb2 = reinterpret_cast<float*>(lds_raw + (size_t)p.C_hid * (CIN + COUT) * 2);  // [C_hid]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kb = lane >> 4;
  const bool folded = p.ab == nullptr;
  const int n_w2 = p.C_hid * CIN / 8, n_w3 = COUT * p.C_hid / 8;                     // 16-byte pieces
  for (int i = tid; i < n_w3; i += NWAVES * 64) lw3[i] = p.w3[i];
  const long Ts = (p.rps + NT * 16 - 1) / (NT * 16);       // row tiles per sample
  const long G = Ts * p.N;
  // this workgroup's contiguous share of the (sample, tile) sequence; its waves take the tiles of a share round-robin
  const long g_end = G * (blockIdx.x + 1) / gridDim.x;
  long g = G * blockIdx.x / gridDim.x;
  const bool with_res = p.e.res_mode != PYTC_RES_NONE;
  const bool ups = p.e.res_mode == PYTC_RES_UPSAMPLE;
  int staged = -1;

  while (g < g_end) {                                      // one pass per sample the share touches (workgroup-uniform)
    const int n = (int)(g / Ts);
    const long seg_end = min(g_end, (long)(n + 1) * Ts);
    if (staged < 0 || folded) {
      __syncthreads();                                     // every wave is done with the previous sample's operands
      const bf16x8_t* src = p.w2 + (folded ? (long)n * p.w2_stride : 0L);
      for (int i = tid; i < n_w2; i += NWAVES * 64) lw2[i] = src[i];
      const float* b2 = p.b2 + (folded ? (long)n * p.C_hid : 0L);
      for (int i = tid; i < p.C_hid; i += NWAVES * 64) lb2[i] = b2[i];
      staged = n;
      __syncthreads();
    }
    const float* an = folded ? nullptr : p.ab + (long)n * 2 * CIN;
    const long t_base = (long)n * Ts;

    for (long gt = g + wave; gt < seg_end; gt += NWAVES) {
      const long row0 = (gt - t_base) * (NT * 16);
      LdsTile<KS_IN, MO, NT> cur;
      lds_tile_load(cur, p, n, row0, r, kb, with_res);
      long orow[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) orow[nt] = row0 + nt * 16 + r;

      // ---- B operand of GEMM1 (pw_mlp_kernel's prologue)
      bf16x8_t bact[KS_IN][NT];
#pragma unroll
      for (int ks = 0; ks < KS_IN; ++ks) {
        if (folded) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bact[ks][nt] = __builtin_bit_cast(bf16x8_t, cur.raw[ks][nt]);
        } else {
          const int k0 = ks * 32 + kb * 8;
          float av[8], bv[8];
          VecIO<float, 4>::load(an + k0, reinterpret_cast<float(&)[4]>(av[0]));
          VecIO<float, 4>::load(an + k0 + 4, reinterpret_cast<float(&)[4]>(av[4]));
          VecIO<float, 4>::load(an + CIN + k0, reinterpret_cast<float(&)[4]>(bv[0]));
          VecIO<float, 4>::load(an + CIN + k0 + 4, reinterpret_cast<float(&)[4]>(bv[4]));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float v[8];
            VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&cur.raw[ks][nt]), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], av[j], bv[j]);
            bact[ks][nt] = Mma<bf16_t>::from_floats(v);
          }
        }
      }
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
      // ---- hidden chunks: GEMM1 -> packed-fp16 GELU -> GEMM2, weight fragments and biases from LDS
      for (int hc = 0; hc < p.HC; ++hc) {
        const f32x4_t b2lo = *reinterpret_cast<const f32x4_t*>(lb2 + hc * 32 + kb * 8);
        const f32x4_t b2hi = *reinterpret_cast<const f32x4_t*>(lb2 + hc * 32 + kb * 8 + 4);
        f32x4_t acc1[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc1[0][nt] = b2lo; acc1[1][nt] = b2hi; }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
          for (int ks = 0; ks < KS_IN; ++ks) {
            const bf16x8_t a = lw2[((hc * 2 + mt) * KS_IN + ks) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc1[mt][nt] = Mma<bf16_t>::mma(a, bact[ks][nt], acc1[mt][nt]);
          }
        }
        h8_t bhh[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float gg[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { gg[j] = acc1[0][nt][j]; gg[4 + j] = acc1[1][nt][j]; }
          bhh[nt] = gelu_h8_from_f32(gg);
        }
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
          const h8_t a = lw3[(mo * p.HC + hc) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc2[mo][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhh[nt], acc2[mo][nt], 0, 0, 0);
        }
      }
      // ---- epilogue (pw_mlp_kernel's)
      int upos[NT][3];
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
          for (int j = 0; j < 4; ++j) { v[j] = acc2[2 * pr][nt][j]; v[4 + j] = acc2[2 * pr + 1][nt][j]; }
          if (with_res) {
            float pre[8];
            VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&cur.res[pr][nt]), pre);
            finish_and_store<bf16_t, 8, false, COUT>(v, p.e, n, orow[nt], pr * 32 + kb * 8, pre, ups ? upos[nt] : nullptr);
          } else {
            finish_and_store<bf16_t, 8, false, COUT>(v, p.e, n, orow[nt], pr * 32 + kb * 8, nullptr, nullptr);
          }
        }
      }
    }
    g = seg_end;
  }
}

template <int KS_IN, int MO, int NT, int NWAVES, int WPS>
static void launch_mlp_lds_v(const MlpLdsParams& p, size_t lds, hipStream_t s) {
  auto kern = &pw_mlp_lds_kernel<KS_IN, MO, NT, NWAVES, WPS>;
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, "pw_mlp_lds")) return;     // per (kernel, device); thread safe
  const long tiles = ((p.rps + NT * 16 - 1) / (NT * 16)) * p.N;
  const int by_lds = (int)((160 * 1024) / lds), by_waves = WPS * 4 / NWAVES;
  const int per_cu = by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves;
  long blocks = 256L * per_cu;
  const long need = (tiles + NWAVES - 1) / NWAVES;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NWAVES * 64), lds, s, p);
}

// variant (knob mlp_lds_variant; 0 = the measured best of the shape): 1 = 8 waves compiled for 4 per SIMD (two workgroups per CU
// when the images fit), 2 = 16 waves (4 per SIMD), 3 = 8 waves at 2 per SIMD, 4 = 12 waves (3 per SIMD).  MI355X, 8 windows
// (profiles/r04_lds_resident_mixer.txt): 64->128->64 at 56^3: 4 (118 us; streaming kernel 150 ... 180); 128->256->64 at 56^3: 4
// (245 us; 385); 128->256->128 at 28^3: 3 (49 us; 59); 64->128->32 at 112^3: 2 (763 us; 820).  The 4-per-SIMD builds of the two
// wide shapes spill.
template <int KS_IN, int MO, int NT>
static void launch_mlp_lds(const MlpLdsParams& p, int variant, hipStream_t s) {
  const size_t lds = (size_t)p.C_hid * (p.C_in + p.C_out) * 2 + (size_t)p.C_hid * 4;
  if (variant == 0) variant = (KS_IN == 4 && MO == 8) ? 3 : (MO == 2 ? 2 : 4);
  switch (variant) {
    case 1: launch_mlp_lds_v<KS_IN, MO, NT, 8, 4>(p, lds, s); break;
    case 2: launch_mlp_lds_v<KS_IN, MO, NT, 16, 4>(p, lds, s); break;
    case 3: launch_mlp_lds_v<KS_IN, MO, NT, 8, 2>(p, lds, s); break;
    default: launch_mlp_lds_v<KS_IN, MO, NT, 12, 3>(p, lds, s); break;
  }
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_pw_mlp_lds_supported(int C_in, int C_hid, int C_out) {
  if (C_in % 32 || C_hid % 32 || C_out % 32) return 0;
  const int ks = C_in / 32, mo = C_out / 16;
  const bool shape = (ks == 2 && mo == 2) || (ks == 2 && mo == 4) || (ks == 4 && mo == 4) || (ks == 4 && mo == 8);
  const size_t lds = (size_t)C_hid * (C_in + C_out) * 2 + (size_t)C_hid * 4;
  return (shape && lds <= 160 * 1024) ? 1 : 0;
}

// pytc_pw_mlp_fwd's arguments and results (w3_format must be PYTC_W3_F16); the mid-level shapes of pytc_pw_mlp_lds_supported.
extern "C" int pytc_pw_mlp_lds_fwd(const pytc_mlp_args* a, void* stream) {
  PYTC_REQUIRE(a && a->t && (a->ab || a->per_sample) && a->w2_packed && a->w3_packed && a->b2 && a->b3 && a->y, "pw_mlp_lds: null pointer");
  PYTC_REQUIRE(!(a->ab && a->per_sample), "pw_mlp_lds: per-sample (norm-folded) expand operands come without an affine");
  PYTC_REQUIRE(a->w3_format == PYTC_W3_F16, "pw_mlp_lds: the projection image must be fp16 (pytc_pw_pack_weight_paired_f16)");
  PYTC_REQUIRE(a->N >= 1 && a->rows_per_sample >= 1, "pw_mlp_lds: bad shape");
  if (!pytc_pw_mlp_lds_supported(a->C_in, a->C_hid, a->C_out)) {
    set_error("pw_mlp_lds: no LDS-resident kernel for C_in=%d C_hid=%d C_out=%d", a->C_in, a->C_hid, a->C_out);
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || a->res, "pw_mlp_lds: residual mode without residual pointer");
  MlpLdsParams p{};
  p.t = (const bf16_t*)a->t; p.ab = a->ab; p.w2 = (const bf16x8_t*)a->w2_packed; p.b2 = a->b2;
  p.w3 = (const h8_t*)a->w3_packed; p.b3 = a->b3;
  p.rps = a->rows_per_sample; p.N = a->N; p.C_in = a->C_in; p.C_hid = a->C_hid; p.C_out = a->C_out; p.HC = a->C_hid / 32;
  p.w2_stride = (long)(a->C_hid / 16) * (a->C_in / 32) * 64;
  p.e.res = a->res; p.e.res_low = a->res_low; p.e.res_bias = a->res_bias; p.e.y = a->y;
  p.e.rps_out = a->rows_per_sample; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode;
  p.e.nt = stream_nt_policy((long)a->N * a->rows_per_sample * (a->C_in > a->C_out ? a->C_in : a->C_out) * 2);
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  if (a->res_mode == PYTC_RES_UPSAMPLE) {
    PYTC_REQUIRE((long)a->Di * a->Hi * a->Wi == a->rows_per_sample && !(a->Di & 1) && !(a->Hi & 1) && !(a->Wi & 1) &&
                 a->rows_per_sample < (1L << 31), "pw_mlp_lds: RES_UPSAMPLE needs the (even) output grid");
    p.e.Go_d = a->Di; p.e.Go_h = a->Hi; p.e.Go_w = a->Wi;
    p.e.Gl_d = a->Di / 2; p.e.Gl_h = a->Hi / 2; p.e.Gl_w = a->Wi / 2;
  }
  hipStream_t s = (hipStream_t)stream;
  const int ks = a->C_in / 32, mo = a->C_out / 16;
  const int variant = tuning_get("mlp_lds_variant", 0);
  if (ks == 2 && mo == 2) launch_mlp_lds<2, 2, 2>(p, variant, s);     // 64 rows per wave (NT = 4): 748 us at best against 735
  else if (ks == 2 && mo == 4) launch_mlp_lds<2, 4, 2>(p, variant, s);
  else if (ks == 4 && mo == 4) launch_mlp_lds<4, 4, 2>(p, variant, s);
  else launch_mlp_lds<4, 8, 2>(p, variant, s);
  PYTC_LAUNCH_CHECK("pw_mlp_lds");
  return PYTC_OK;
}
