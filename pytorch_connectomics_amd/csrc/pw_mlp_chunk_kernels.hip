// Fused MedNeXt channel mixer with the weights STREAMED THROUGH LDS, one hidden chunk at a time (round 5) -- wide hidden layers whose
// images do not fit LDS (MedNeXt-L: 128->1024->128 at 40^3, 256->2048->128, 128->512->64, 64->512->128).
//
// pw_mlp_kernel streams every weight fragment of every hidden chunk from L2 per WAVE: a wave of 32 rows pulls the whole image pair
// (128->1024->128: 512 KB) through the vector memory pipe, 16 waves per CU -- 2 GB of L2 -> CU traffic per launch of 128 000 rows at the
// 64 B / clk a CU takes in, 148 us where the MFMAs need 27 (profiles/r05_mednext_l_labels_after.txt).  pw_mlp_lds_kernel keeps both images
// resident in LDS, which these shapes exceed.  Here the NW waves of a workgroup (32 rows each) share every 32-wide hidden chunk: its
// fragments -- 2 * KS_IN KB of the expand image, MO KB of the projection image -- travel L2 -> LDS ONCE per workgroup by
// `global_load_lds_dwordx4` (1 KB pieces dealt over the waves, no registers), two chunks ahead of the chunk being computed, into a
// ring of three buffers; one workgroup barrier per chunk (it publishes chunk k and retires the buffer of chunk k - 1, which chunk k + 2
// then overwrites).  Fragments are ds_read_b128 at lane-contiguous addresses (conflict-free).  L2 -> CU weight traffic drops NW-fold.
// The arithmetic is pw_mlp_kernel<KS_IN, MO, 2, GELU_MODE = 3>'s, instruction for instruction (same MFMA order, packed-fp16 GELU,
// epilogue): results are bit-identical (tests/test_gpu_kernels.py::test_chunk_streamed_mixer_is_bit_identical).
//
// vmcnt discipline: the DMA pieces are asm-issued (hipcc does not count them).  Each wave waits for ITS pieces of chunk k with
// `s_waitcnt vmcnt(P)` (P = its pieces per chunk: those of chunk k + 1 may still fly; loads return in order), then the barrier makes
// every wave's pieces visible.  hipcc's own loads (activation rows before the loop, residual rows after it) only ever see additional
// YOUNGER or OLDER asm loads in the counter, which makes its counted waits conservative, never short.  No stores inside the loop.
#include "pw_common.h"

namespace pytc {

struct MlpChunkParams {
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

__device__ __forceinline__ void chunk_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int KS_IN, int MO, int NW, int WPS>
__global__ void __launch_bounds__(NW * 64, WPS)
pw_mlp_chunk_kernel(MlpChunkParams p) {
  constexpr int NT = 2, CIN = KS_IN * 32, COUT = MO * 16;
  constexpr int PIECES = 2 * KS_IN + MO;                   // 1 KB pieces per chunk: expand fragments [mt][ks], then projection fragments [mo]
  constexpr int MINE = (PIECES + NW - 1) / NW;             // pieces a wave may own (wave w: w, w + NW, ...)
  constexpr int NBUF = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char chunk_lds[];
  uint4* ring = reinterpret_cast<uint4*>(chunk_lds);                                  // [NBUF][PIECES][64 lanes]
  float* lb2 = reinterpret_cast<float*>(chunk_lds + (size_t)NBUF * PIECES * 1024);    // [C_hid]
  float* lb3 = lb2 + p.C_hid;                                                          // [COUT]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform: it selects DMA pieces and LDS bases (SGPR operands)
  const int r = lane & 15, kb = lane >> 4;
  const int n = blockIdx.y;
  const bool folded = p.ab == nullptr;
  const long row0 = ((long)blockIdx.x * NW + wave) * (NT * 16);
  const bool with_res = p.e.res_mode != PYTC_RES_NONE;
  const bool ups = p.e.res_mode == PYTC_RES_UPSAMPLE;

  // ---- the pieces this wave moves per chunk (wave-uniform), and their DMA
  const unsigned char* w2img = reinterpret_cast<const unsigned char*>(p.w2 + (folded ? (long)n * p.w2_stride : 0L));
  const unsigned char* w3img = reinterpret_cast<const unsigned char*>(p.w3);
  const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ring);
  int mine = 0;
#pragma unroll
  for (int i = 0; i < MINE; ++i) mine += (wave + i * NW < PIECES) ? 1 : 0;
  auto request = [&](int hc) {
    const unsigned buf = lbase + (unsigned)((hc % NBUF) * PIECES) * 1024u;
#pragma unroll
    for (int i = 0; i < MINE; ++i) {
      const int pc = wave + i * NW;                          // wave-uniform
      if (pc < PIECES) {
        const unsigned char* src = pc < 2 * KS_IN ? w2img + ((long)hc * (2 * KS_IN) + pc) * 1024
                                                  : w3img + ((long)(pc - 2 * KS_IN) * p.HC + hc) * 1024;
        chunk_glds16(src + lane * 16, buf + (unsigned)pc * 1024u);
      }
    }
  };

  // ---- B operand of GEMM1 (pw_mlp_kernel's prologue): requested first, the first two chunks' DMA right behind them
  long orow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) orow[nt] = row0 + nt * 16 + r;
  bf16x8_t bact[KS_IN][NT];
  const bf16_t* tn = p.t + (long)n * p.rps * CIN;
  uint4 raw[KS_IN][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const long rr = orow[nt] < p.rps ? orow[nt] : p.rps - 1;
#pragma unroll
    for (int ks = 0; ks < KS_IN; ++ks) raw[ks][nt] = ld_stream(reinterpret_cast<const uint4*>(tn + rr * CIN + ks * 32 + kb * 8), p.e.nt);
  }
  {
    const float* b2 = p.b2 + (folded ? (long)n * p.C_hid : 0L);
    for (int i = tid; i < p.C_hid; i += NW * 64) lb2[i] = b2[i];
    if (tid < COUT) lb3[tid] = p.b3[tid];
  }
  request(0);
  if (p.HC > 1) request(1);
  const float* an = folded ? nullptr : p.ab + (long)n * 2 * CIN;
#pragma unroll
  for (int ks = 0; ks < KS_IN; ++ks) {
    if (folded) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bact[ks][nt] = __builtin_bit_cast(bf16x8_t, raw[ks][nt]);
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
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&raw[ks][nt]), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], av[j], bv[j]);
        bact[ks][nt] = Mma<bf16_t>::from_floats(v);
      }
    }
  }
  // hipcc would otherwise place its waits for the activation rows at their first use INSIDE the chunk loop, where a `vmcnt(0)` also waits for the
  // chunk requested a moment before (every iteration): consume them here
#pragma unroll
  for (int ks = 0; ks < KS_IN; ++ks)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(bact[ks][nt]));
  __syncthreads();                                           // lb2 / lb3 are staged
  f32x4_t acc2[MO][NT];
#pragma unroll
  for (int pr = 0; pr < MO / 2; ++pr) {
    const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(lb3 + pr * 32 + kb * 8), hi = *reinterpret_cast<const f32x4_t*>(lb3 + pr * 32 + kb * 8 + 4);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc2[2 * pr][nt] = lo; acc2[2 * pr + 1][nt] = hi; }
  }

  // ---- hidden chunks: GEMM1 -> packed-fp16 GELU -> GEMM2, fragments from the ring
  for (int hc = 0; hc < p.HC; ++hc) {
    // this wave's pieces of chunk hc have landed (those of chunk hc + 1 may still be in flight) ...
    if (hc + 1 < p.HC) {
      if (mine == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (mine == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (mine == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                         // ... and everybody's; every wave is done with chunk hc - 1
    if (hc + 2 < p.HC) request(hc + 2);                      // into the buffer chunk hc - 1 occupied
    const uint4* cb = ring + (size_t)(hc % NBUF) * PIECES * 64;
    const f32x4_t b2lo = *reinterpret_cast<const f32x4_t*>(lb2 + hc * 32 + kb * 8);
    const f32x4_t b2hi = *reinterpret_cast<const f32x4_t*>(lb2 + hc * 32 + kb * 8 + 4);
    f32x4_t acc1[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc1[0][nt] = b2lo; acc1[1][nt] = b2hi; }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int ks = 0; ks < KS_IN; ++ks) {
        const bf16x8_t a = __builtin_bit_cast(bf16x8_t, cb[(mt * KS_IN + ks) * 64 + lane]);
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
      const h8_t a = __builtin_bit_cast(h8_t, cb[(2 * KS_IN + mo) * 64 + lane]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc2[mo][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhh[nt], acc2[mo][nt], 0, 0, 0);
    }
  }

  // ---- epilogue (pw_mlp_kernel's)
  if (row0 >= p.rps) return;
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
      finish_and_store<bf16_t, 8, false, COUT>(v, p.e, n, orow[nt], pr * 32 + kb * 8, nullptr, (with_res && ups) ? upos[nt] : nullptr);
    }
  }
}

template <int KS_IN, int MO, int NW, int WPS>
static void launch_mlp_chunk_v(const MlpChunkParams& p, hipStream_t s) {
  auto kern = &pw_mlp_chunk_kernel<KS_IN, MO, NW, WPS>;
  const size_t lds = (size_t)3 * (2 * KS_IN + MO) * 1024 + (size_t)(p.C_hid + MO * 16) * 4;
  // the opt-in is remembered per (kernel, device): ask for the device's whole LDS once, not for this launch's size (a later launch of the same
  // instance with a wider hidden layer -- 256 -> 512 -> 128, then 256 -> 2048 -> 128 -- needs more for its bias vector)
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, "pw_mlp_chunk")) return;     // thread safe
  const long wg_rows = (long)NW * 32;
  dim3 grid((unsigned)((p.rps + wg_rows - 1) / wg_rows), (unsigned)p.N), block(NW * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, s, p);
}

// variant (knob mlp_chunk_variant; 0 = the measured best of the shape): 1 = 8 waves at 2 per SIMD, 2 = 12 waves at 3 per SIMD,
// 3 = 8 waves compiled for 4 per SIMD (two workgroups per CU), 4 = 16 waves at 4 per SIMD
template <int KS_IN, int MO>
static void launch_mlp_chunk(const MlpChunkParams& p, int variant, hipStream_t s) {
  // MI355X (profiles/r05_chunk_streamed_mixer.txt): two 8-wave workgroups per CU win wherever the registers allow 4 waves per SIMD
  if (variant == 0) variant = (KS_IN == 8) ? 1 : 3;
  switch (variant) {
    case 2: launch_mlp_chunk_v<KS_IN, MO, 12, 3>(p, s); break;
    case 3: launch_mlp_chunk_v<KS_IN, MO, 8, 4>(p, s); break;
    case 4: launch_mlp_chunk_v<KS_IN, MO, 16, 4>(p, s); break;
    default: launch_mlp_chunk_v<KS_IN, MO, 8, 2>(p, s); break;
  }
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_pw_mlp_chunk_supported(int C_in, int C_hid, int C_out) {
  if (C_in % 32 || C_hid % 32 || C_out % 32) return 0;
  const int ks = C_in / 32, mo = C_out / 16;
  const bool shape = (ks == 2 && (mo == 2 || mo == 4 || mo == 8)) || (ks == 4 && mo == 4) || (ks == 4 && mo == 8) || (ks == 8 && mo == 8);
  return (shape && C_hid >= 64 && C_hid <= 8192) ? 1 : 0;
}

// pytc_pw_mlp_fwd's arguments and results (w3_format must be PYTC_W3_F16); the shapes of pytc_pw_mlp_chunk_supported.
extern "C" int pytc_pw_mlp_chunk_fwd(const pytc_mlp_args* a, void* stream) {
  PYTC_REQUIRE(a && a->t && (a->ab || a->per_sample) && a->w2_packed && a->w3_packed && a->b2 && a->b3 && a->y, "pw_mlp_chunk: null pointer");
  PYTC_REQUIRE(!(a->ab && a->per_sample), "pw_mlp_chunk: per-sample (norm-folded) expand operands come without an affine");
  PYTC_REQUIRE(a->w3_format == PYTC_W3_F16, "pw_mlp_chunk: the projection image must be fp16 (pytc_pw_pack_weight_paired_f16)");
  PYTC_REQUIRE(a->N >= 1 && a->rows_per_sample >= 1, "pw_mlp_chunk: bad shape");
  if (!pytc_pw_mlp_chunk_supported(a->C_in, a->C_hid, a->C_out)) {
    set_error("pw_mlp_chunk: no chunk-streamed kernel for C_in=%d C_hid=%d C_out=%d", a->C_in, a->C_hid, a->C_out);
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || a->res, "pw_mlp_chunk: residual mode without residual pointer");
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || a->res_mode == PYTC_RES_ADD || a->res_mode == PYTC_RES_UPSAMPLE, "pw_mlp_chunk: residual mode");
  MlpChunkParams p{};
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
                 a->rows_per_sample < (1L << 31), "pw_mlp_chunk: RES_UPSAMPLE needs the (even) output grid");
    p.e.Go_d = a->Di; p.e.Go_h = a->Hi; p.e.Go_w = a->Wi;
    p.e.Gl_d = a->Di / 2; p.e.Gl_h = a->Hi / 2; p.e.Gl_w = a->Wi / 2;
  }
  hipStream_t s = (hipStream_t)stream;
  const int ks = a->C_in / 32, mo = a->C_out / 16;
  const int variant = tuning_get("mlp_chunk_variant", 0);
  if (ks == 2 && mo == 2) launch_mlp_chunk<2, 2>(p, variant, s);
  else if (ks == 2 && mo == 4) launch_mlp_chunk<2, 4>(p, variant, s);
  else if (ks == 2 && mo == 8) launch_mlp_chunk<2, 8>(p, variant, s);
  else if (ks == 4 && mo == 4) launch_mlp_chunk<4, 4>(p, variant, s);
  else if (ks == 4 && mo == 8) launch_mlp_chunk<4, 8>(p, variant, s);
  else launch_mlp_chunk<8, 8>(p, variant, s);
  PYTC_LAUNCH_CHECK("pw_mlp_chunk");
  return PYTC_OK;
}
