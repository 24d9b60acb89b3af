// LDS-tiled MFMA GEMM for the 1x1x1 convs of the DEEP levels (round 4):  Y[r][o] = epi( sum_k f(X[r][k]) * W[o][k] + b[o] )
//
// The fused mixer (pw_mlp_kernels.hip) gives every wave 16 voxel rows and lets it walk the whole hidden dimension: right where
// rows are plentiful (levels 0-2: the expanded tensor never leaves registers), wrong below ~30 k rows -- at 14^3 / 7^3 voxels per
// window 172 ... 1372 waves exist for 1024 SIMDs, each streams EVERY weight fragment from L2 for its 16 rows, and the launch runs
// at 3 % of the MFMA peak (512 -> 1024 -> 512 on 2 744 rows: 82 us).  Here the two convs of a deep block are two launches of a
// conventional GEMM: a workgroup owns BR rows x 128 output channels, both operands are staged through double-buffered LDS in
// 64-wide k steps (global loads of step k+1 in flight during the MFMAs of step k), every weight element is read once per
// 64 / 128 rows instead of once per 16, and the grid has (rows / BR) x (C_out / 128) workgroups.  The hidden tensor makes one round
// trip through L2 / Infinity Cache (5.6 MB at the bottleneck of an 8-window batch).
//
// Arithmetic is the fused mixer's, operation for operation: weights = MFMA A operand with the paired-row permutation (a lane ends with
// 8 consecutive output channels), accumulators start from the bias, k ascends in steps of 32, the GroupNorm affine is applied in
// fp32 and rounded to bf16 while staging, the hidden activation is the packed-fp16 polynomial GELU stored as fp16 and the
// projection runs on v_mfma_f32_16x16x32_f16 -- so a block computed this way is BIT-IDENTICAL to the fused kernel
// (tests/test_gpu_kernels.py::test_deep_level_gemm_pair_is_bit_identical_to_the_fused_mixer).
#include "pw_common.h"

namespace pytc {

struct GemmParams {
  const unsigned short* x;    // [N * rps][C_in], bf16 (GEMM 1) or fp16 (GEMM 2)
  const unsigned short* w;    // [C_out][C_in] row-major, same 16-bit type as x
  const float* bias;          // [C_out]
  const float* ab;            // [N][2][C_in] GroupNorm affine applied to x while staging, or null
  unsigned short* yh;         // GELU epilogue: fp16 [N * rps][C_out]
  EpiParams e;                // plain epilogue: bf16 output with the residual variants of finish_and_store
  long rps, rows_total;
  int C_in, C_out, gelu;
  int pre_gelu;               // bf16 operand: x <- bf16(gelu_fast(x)) while staging (the training forward's projecting conv reads the
                              // stored pre-activation; the same function pw_fast_kernel applies and the weight-gradient kernels recompute)
};

constexpr int GEMM_BN = 128;            // output channels per workgroup
constexpr int GEMM_KS = 64;             // k per staged step = two 16x16x32 MFMA sub-steps (one barrier per 64 k); the template's default

// KS = 32 (round 5): half the LDS per workgroup (41 KB at BR = 128, 31 KB at BR = 64) -> 3 / 4 workgroups per CU instead of 2.  These launches
// are latency bound, not MFMA bound (256 -> 2048 on 16 000 rows: a workgroup lives 12 us for 0.85 us of MFMAs -- every k step exposes a global
// load round trip that one step of compute cannot cover), so resident workgroups are what hides it.  Same k order: bit-identical.
// BWD (round 6: the deep-level GEMMs of the TRAINING step -- expand with the stored bf16 pre-activation as output, project with the GELU
// in its operand prologue, and the two data-gradient GEMMs): the epilogue keeps finish_and_store's PYTC_RES_GELU_BWD / PYTC_RES_NORM_BWD
// branches, the bias may be null.  pw_fast_kernel, which served these launches, keeps a wave's whole operand tile in registers and streams
// EVERY weight fragment from L2 per 16-64 rows: 1024 -> 512 on 1 372 rows ran 43 us = 33 TFLOP/s.
template <bool F16, int BR, int KS = GEMM_KS, bool BWD = false>     // F16: fp16 operands (the projecting conv); BR rows per workgroup (64 or 128)
__global__ void __launch_bounds__(256, KS == 32 ? (BR == 64 ? 4 : 3) : 2)
pw_gemm_lds_kernel(GemmParams p) {
  constexpr int GEMM_KS = KS;
  constexpr int GEMM_PITCH = GEMM_KS + 8; // LDS row pitch in elements (144 / 80 bytes: the 16 rows of a fragment read hit 16 distinct 16-byte columns)
  constexpr int NT = BR / 2 / 16;       // 16-row tiles per wave (2 row halves per workgroup)
  constexpr int MT = 4;                 // 16-channel tiles per wave (2 channel halves of 64)
  constexpr int CPR = GEMM_KS / 8;      // 16-byte pieces per staged row
  constexpr int A_CHUNKS = GEMM_BN * CPR / 256, B_CHUNKS = BR * CPR / 256;     // pieces per thread and k step
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][GEMM_BN * GEMM_PITCH];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][BR * GEMM_PITCH];
  typedef unsigned int q4_t __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int r = lane & 15, kb = lane >> 4;
  // 1-D grid, XCD-aware: logical index = row tile * (column tiles) + column tile, and consecutive logical indices run on ONE XCD in
  // dispatch order -- the C_out / 128 workgroups that read the same 64 / 128 activation rows are neighbours in time on one L2, so a
  // row tile comes from HBM / Infinity Cache once (first cut: grid (row tiles, column tiles), the sharers 172 dispatches apart on
  // eight different XCDs -- every activation tile fetched 4-8 times, the launch bound by that traffic at 5.5 TB/s)
  const int tiles_c = p.C_out / GEMM_BN;
  const int lb_ = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const long r0 = (long)(lb_ / tiles_c) * BR;
  const int o0 = (lb_ % tiles_c) * GEMM_BN;
  const bool affine = !F16 && p.ab != nullptr;              // workgroup-uniform

  // ---- staging descriptors: thread -> (row, 16-byte piece) of each operand tile, constant over k
  const unsigned short* ga[A_CHUNKS];
  const unsigned short* gb[B_CHUNKS];
  int la[A_CHUNKS], lb[B_CHUNKS];
  const float* abn[B_CHUNKS];
#pragma unroll
  for (int i = 0; i < A_CHUNKS; ++i) {
    const int c = tid + 256 * i, row = c / CPR, piece = c % CPR;
    ga[i] = p.w + (long)(o0 + row) * p.C_in + piece * 8;
    la[i] = row * GEMM_PITCH + piece * 8;
  }
#pragma unroll
  for (int i = 0; i < B_CHUNKS; ++i) {
    const int c = tid + 256 * i, row = c / CPR, piece = c % CPR;
    long gr = r0 + row;
    if (gr >= p.rows_total) gr = p.rows_total - 1;          // clamped: results of rows beyond the end are never stored
    gb[i] = p.x + gr * p.C_in + piece * 8;
    lb[i] = row * GEMM_PITCH + piece * 8;
    abn[i] = affine ? p.ab + (gr / p.rps) * 2 * p.C_in + piece * 8 : nullptr;
  }
  // one register stage: the loads of k step kt + 1 -- operand pieces AND the affine rows they meet -- are issued before the MFMAs of
  // step kt (at the deep levels a launch has about one workgroup per CU: an L2 / Infinity-Cache round trip has to be covered inside
  // the workgroup; the first cut loaded the affine inside `commit` and paid that round trip once per step: 0.8 us per 32 k)
  q4_t ra[A_CHUNKS], rb[B_CHUNKS];
  f32x4_t rab[F16 ? 1 : B_CHUNKS][4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) ra[i] = *reinterpret_cast<const q4_t*>(ga[i] + k0);
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) rb[i] = *reinterpret_cast<const q4_t*>(gb[i] + k0);
    if constexpr (!F16) {
      if (affine) {
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
          rab[i][0] = *reinterpret_cast<const f32x4_t*>(abn[i] + k0);
          rab[i][1] = *reinterpret_cast<const f32x4_t*>(abn[i] + k0 + 4);
          rab[i][2] = *reinterpret_cast<const f32x4_t*>(abn[i] + p.C_in + k0);
          rab[i][3] = *reinterpret_cast<const f32x4_t*>(abn[i] + p.C_in + k0 + 4);
        }
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) *reinterpret_cast<q4_t*>(&sA[buf][la[i]]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      q4_t v = rb[i];
      if constexpr (!F16) {
        if (affine || (BWD && p.pre_gelu)) {   // GroupNorm affine in fp32, rounded to bf16: what the fused mixer's prologue does
          float f[8];
          VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&v), f);
          if (affine) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f[j] = fmaf(f[j], rab[i][0][j], rab[i][2][j]);
              f[4 + j] = fmaf(f[4 + j], rab[i][1][j], rab[i][3][j]);
            }
          }
          if constexpr (BWD) {
            if (p.pre_gelu) {
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] = gelu_fast(f[j]);
            }
          }
          v = __builtin_bit_cast(q4_t, Mma<bf16_t>::from_floats(f));
        }
      }
      *reinterpret_cast<q4_t*>(&sB[buf][lb[i]]) = v;
    }
  };

  // ---- accumulators start from the bias (8 consecutive channels per lane and tile pair, like the fused mixer)
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int pr = 0; pr < MT / 2; ++pr) {
    float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ch = o0 + wm * 64 + pr * 32 + kb * 8;
    if (!BWD || p.bias) {
      VecIO<float, 4>::load(p.bias + ch, reinterpret_cast<float(&)[4]>(b[0]));
      VecIO<float, 4>::load(p.bias + ch + 4, reinterpret_cast<float(&)[4]>(b[4]));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[2 * pr][nt] = f32x4_t{b[0], b[1], b[2], b[3]};
      acc[2 * pr + 1][nt] = f32x4_t{b[4], b[5], b[6], b[7]};
    }
  }
  // fragment addresses: A tile T row m <-> weight row paired_row(T, m) of the wave's 64 channels; B tile nt row r
  int fa[MT], fb[NT];
#pragma unroll
  for (int t = 0; t < MT; ++t) fa[t] = (wm * 64 + paired_row(t, r)) * GEMM_PITCH + kb * 8;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) fb[nt] = (wn * (BR / 2) + nt * 16 + r) * GEMM_PITCH + kb * 8;

  const int KT = p.C_in / GEMM_KS;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) fetch((kt + 1) * GEMM_KS);
#pragma unroll
    for (int sub = 0; sub < GEMM_KS / 32; ++sub) {          // k ascending in 32-wide MFMA steps: the fused mixer's summation order
      bf16x8_t af[MT], bfr[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) af[t] = *reinterpret_cast<const bf16x8_t*>(&sA[buf][fa[t] + sub * 32]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bfr[nt] = *reinterpret_cast<const bf16x8_t*>(&sB[buf][fb[nt] + sub * 32]);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (F16)
            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, af[t]), __builtin_bit_cast(h8_t, bfr[nt]),
                                                                acc[t][nt], 0, 0, 0);
          else
            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bfr[nt], acc[t][nt], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) commit(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: 8 consecutive output channels of one row per lane and tile pair
#pragma unroll
  for (int pr = 0; pr < MT / 2; ++pr) {
    const int ch = o0 + wm * 64 + pr * 32 + kb * 8;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const long row = r0 + wn * (BR / 2) + nt * 16 + r;
      if (row >= p.rows_total) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc[2 * pr][nt][j]; v[4 + j] = acc[2 * pr + 1][nt][j]; }
      if (p.gelu) {
        const h8_t g = gelu_h8_from_f32(v);
        *reinterpret_cast<h8_t*>(p.yh + row * p.C_out + ch) = g;
      } else {
        const int n = (int)(row / p.rps);
        finish_and_store<bf16_t, 8, BWD>(v, p.e, n, row - (long)n * p.rps, ch);
      }
    }
  }
}

// pytc_pw_conv_fwd with w_paired == 2: the weight is a plain row-major bf16 [C_out][C_in] matrix and the launch is this kernel (bf16 in / out)
bool pw_gemm_rowmajor_supported(const pytc_pw_args* a) {
  return a->in_dtype == PYTC_BF16 && a->out_dtype == PYTC_BF16 && a->w_dtype == PYTC_BF16 && a->C_in % GEMM_KS == 0 && a->C_in >= GEMM_KS &&
         a->C_out % GEMM_BN == 0 && a->C_out >= GEMM_BN && a->gather == 0 && a->act == PYTC_ACT_NONE;
}

void pw_gemm_rowmajor_launch(const pytc_pw_args* a, const EpiParams& e, hipStream_t s) {
  GemmParams p{};
  p.x = (const unsigned short*)a->x; p.w = (const unsigned short*)a->w_packed; p.bias = a->bias; p.ab = a->ab;
  p.yh = nullptr; p.e = e;
  p.rps = a->rows_per_sample; p.rows_total = (long)a->N * a->rows_per_sample; p.C_in = a->C_in; p.C_out = a->C_out; p.gelu = 0;
  p.pre_gelu = a->pre_act == PYTC_ACT_GELU ? 1 : 0;
  const long tiles128 = (p.rows_total + 127) / 128 * (a->C_out / GEMM_BN);
  const bool big = tiles128 >= 512;
  const bool ks32 = a->C_in <= 512;
  if (big) {
    dim3 grid((unsigned)((p.rows_total + 127) / 128 * (a->C_out / GEMM_BN)));
    if (ks32) hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 128, 32, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 128, GEMM_KS, true>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid((unsigned)((p.rows_total + 63) / 64 * (a->C_out / GEMM_BN)));
    if (ks32) hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 64, 32, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 64, GEMM_KS, true>), grid, dim3(256), 0, s, p);
  }
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_pw_gemm_supported(int C_in, int C_out) {
  return (C_in % GEMM_KS == 0 && C_in >= GEMM_KS && C_out % GEMM_BN == 0 && C_out >= GEMM_BN) ? 1 : 0;
}

// x: [N * rows_per_sample][C_in] bf16 (in_f16 = 0) or fp16 (in_f16 = 1); w: [C_out][C_in] row-major in the same 16-bit type;
// gelu_out = 1: y = fp16 [rows][C_out] = gelu(acc) (no residual); gelu_out = 0: y = bf16 with res / res_low / res_bias / res_mode /
// (Di, Hi, Wi) as pytc_pw_mlp_fwd.  ab (bf16 input only): GroupNorm affine [N][2][C_in] applied to x.
extern "C" int pytc_pw_gemm_fwd(const void* x, const void* w, const float* bias, const float* ab, void* y, int N,
                                int64_t rows_per_sample, int C_in, int C_out, int in_f16, int gelu_out, const void* res,
                                const void* res_low, const float* res_bias, int res_mode, int Di, int Hi, int Wi, void* stream) {
  PYTC_REQUIRE(x && w && bias && y && N >= 1 && rows_per_sample >= 1, "pw_gemm: bad arguments");
  PYTC_REQUIRE(pytc_pw_gemm_supported(C_in, C_out), "pw_gemm: C_in=%d must be a multiple of 64 and C_out=%d of 128", C_in, C_out);
  PYTC_REQUIRE(!(in_f16 && ab), "pw_gemm: the GroupNorm affine rides with the bf16 operand");
  PYTC_REQUIRE(!(gelu_out && res_mode != PYTC_RES_NONE), "pw_gemm: the GELU epilogue has no residual");
  PYTC_REQUIRE(res_mode == PYTC_RES_NONE || res, "pw_gemm: residual mode without residual pointer");
  GemmParams p{};
  p.x = (const unsigned short*)x; p.w = (const unsigned short*)w; p.bias = bias; p.ab = ab;
  p.yh = (unsigned short*)y;
  p.rps = rows_per_sample; p.rows_total = (long)N * rows_per_sample; p.C_in = C_in; p.C_out = C_out; p.gelu = gelu_out;
  p.e.res = res; p.e.res_low = res_low; p.e.res_bias = res_bias; p.e.y = y; p.e.rps_out = rows_per_sample; p.e.C_out = C_out;
  p.e.res_mode = res_mode; p.e.nt = 0;
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  if (res_mode == PYTC_RES_UPSAMPLE) {
    PYTC_REQUIRE((long)Di * Hi * Wi == rows_per_sample && !(Di & 1) && !(Hi & 1) && !(Wi & 1) && rows_per_sample < (1L << 31),
                 "pw_gemm: RES_UPSAMPLE needs the (even) output grid");
    p.e.Go_d = Di; p.e.Go_h = Hi; p.e.Go_w = Wi; p.e.Gl_d = Di / 2; p.e.Gl_h = Hi / 2; p.e.Gl_w = Wi / 2;
  }
  hipStream_t s = (hipStream_t)stream;
  // 128-row workgroups once they alone fill the chip twice over, 64-row ones below (the bottleneck level: 2 744 rows)
  const long tiles128 = (p.rows_total + 127) / 128 * (C_out / GEMM_BN);
  const bool big = tiles128 >= 512 && tuning_get("pw_gemm_rows", 0) != 64;
  // 32-wide k steps for the short-K (expanding) GEMMs, 64-wide for the long-K (projecting) ones: MI355X, profiles/r05_gemm_k_step.txt --
  // 256 -> 2048 on 16 000 rows 51 -> 46 us, 512 -> 1024 on 21 952 rows 61 -> 57; 2048 -> 256 34 -> 40 (a barrier per 32 k over 64 steps)
  const int ks_knob = tuning_get("pw_gemm_ks", 0);
  const bool ks32 = ks_knob ? ks_knob == 32 : C_in <= 512;
  if (big || tuning_get("pw_gemm_rows", 0) == 128) {
    dim3 grid((unsigned)((p.rows_total + 127) / 128 * (C_out / GEMM_BN)));
    if (ks32) {
      if (in_f16) hipLaunchKernelGGL((pw_gemm_lds_kernel<true, 128, 32>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 128, 32>), grid, dim3(256), 0, s, p);
    } else {
      if (in_f16) hipLaunchKernelGGL((pw_gemm_lds_kernel<true, 128>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 128>), grid, dim3(256), 0, s, p);
    }
  } else {
    dim3 grid((unsigned)((p.rows_total + 63) / 64 * (C_out / GEMM_BN)));
    if (ks32) {
      if (in_f16) hipLaunchKernelGGL((pw_gemm_lds_kernel<true, 64, 32>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 64, 32>), grid, dim3(256), 0, s, p);
    } else {
      if (in_f16) hipLaunchKernelGGL((pw_gemm_lds_kernel<true, 64>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((pw_gemm_lds_kernel<false, 64>), grid, dim3(256), 0, s, p);
    }
  }
  PYTC_LAUNCH_CHECK("pw_gemm");
  return PYTC_OK;
}
