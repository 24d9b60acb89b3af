// Single pointwise GEMM on the paired-row weight image (bf16 in / weights / out, C_in and C_out multiples of 32):
// the training step's expand / project / data-gradient convs and the un-fused 1x1 convs of the inference path.
// Same transposed form as pw_kernels.hip (weights = MFMA A, voxels = N), with what made pw_mlp fast:
//   * all B-operand loads of a wave's voxel tile (16 B per lane per 32-channel step) are issued up front, the
//     GroupNorm affine and/or GELU pre-activation are applied while they become MFMA operands;
//   * output channels are produced one tile PAIR at a time: after the paired-row packing a lane ends with 8 consecutive
//     channels of one voxel -> 16-byte stores, 16-byte residual / pre-activation loads issued before the pair's MFMAs;
//   * only 2 x NT accumulators live at a time, so the register budget goes to the operand tile (C_in up to 1024).
// HBM traffic = each operand once: C_in + C_out (+ C_out residual) elements per voxel.
#include "pw_common.h"

namespace pytc {

struct PwFastParams {
  const bf16_t* x;
  const bf16x8_t* w;      // paired image [C_out/16][KS][64 lanes][8]
  const float* bias;
  const float* ab;
  EpiParams e;
  long rps, rps_in;       // output / input rows per sample (differ with gather)
  int C_in, C_out, pre_act;
  int gather, Hi, Wi, Ho, Wo;   // gather == 2: output row (oz,oy,ox) reads input voxel (2oz,2oy,2ox)
};

// waves/SIMD the kernel is compiled for (a bare launch_bounds(256) lets hipcc budget for ONE wave per SIMD)
constexpr int pw_fast_waves(int ks, int nt) {
  return nt == 4 ? (ks <= 2 ? 3 : 2) : (ks <= 4 ? 4 : (ks <= 16 ? 3 : 1));
}

template <int KS, int NT>
__global__ void __launch_bounds__(256, pw_fast_waves(KS, NT))
pw_fast_kernel(PwFastParams p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.y;
  constexpr bool WARM = KS >= 4;             // deep levels: warm this XCD's L2 with the weight image (pw_common.h)
  WarmRegs<2> warm;
  if (WARM) {
    const long wbytes = (long)p.C_out * p.C_in * 2;
    warm_l2(p.w, wbytes, warm, 0);
    warm_l2(reinterpret_cast<const char*>(p.w) + (wbytes >> 1), wbytes >> 1, warm, 2);
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

  bf16x8_t bact[KS][NT];
  const bf16_t* xn = p.x + (long)n * p.rps_in * p.C_in;
  {
    uint4 raw[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        long rr = orow[nt] < p.rps ? orow[nt] : p.rps - 1;
        if (p.gather == 2) {
          const int ox = (int)(rr % p.Wo);
          const long t = rr / p.Wo;
          const int oy = (int)(t % p.Ho), oz = (int)(t / p.Ho);
          rr = ((long)(2 * oz) * p.Hi + 2 * oy) * p.Wi + 2 * ox;
        }
        raw[ks][nt] = *reinterpret_cast<const uint4*>(xn + rr * p.C_in + ks * 32 + kb * 8);
      }
    if (p.ab || p.pre_act == PYTC_ACT_GELU) {
      const float* an = p.ab ? p.ab + (long)n * 2 * p.C_in : nullptr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = ks * 32 + kb * 8;
        float av[8], bv[8];
        if (an) {
          VecIO<float, 4>::load(an + k0, reinterpret_cast<float(&)[4]>(av[0]));
          VecIO<float, 4>::load(an + k0 + 4, reinterpret_cast<float(&)[4]>(av[4]));
          VecIO<float, 4>::load(an + p.C_in + k0, reinterpret_cast<float(&)[4]>(bv[0]));
          VecIO<float, 4>::load(an + p.C_in + k0 + 4, reinterpret_cast<float(&)[4]>(bv[4]));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v[8];
          VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&raw[ks][nt]), v);
          if (an) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], av[j], bv[j]);
          }
          if (p.pre_act == PYTC_ACT_GELU) {   // sigmoid-form GELU (|err| <= 2.5e-5, below the bf16 rounding of its result);
#pragma unroll                           // the weight-gradient kernel recomputes the same function
            for (int j = 0; j < 8; ++j) v[j] = gelu_fast(v[j]);
          }
          bact[ks][nt] = Mma<bf16_t>::from_floats(v);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bact[ks][nt] = __builtin_bit_cast(bf16x8_t, raw[ks][nt]);
    }
  }

  if (WARM) warm_l2_done(warm);
  const bool pre_ok = p.e.res_mode == PYTC_RES_ADD || p.e.res_mode == PYTC_RES_GELU_BWD || p.e.res_mode == PYTC_RES_NORM_BWD;
  const bf16_t* resn = pre_ok ? reinterpret_cast<const bf16_t*>(p.e.res) + (long)n * p.rps * p.C_out : nullptr;
  // blockIdx.z owns a contiguous share of the output-channel pairs (launch_fast: > 1 only when the rows alone leave CUs idle)
  const int pairs_z = p.C_out / 32 / (int)gridDim.z;
  for (int pr = blockIdx.z * pairs_z; pr < (int)(blockIdx.z + 1) * pairs_z; ++pr) {
    const int o0 = pr * 32 + kb * 8;
    uint4 rpre[NT];
    if (pre_ok) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const long rr = orow[nt] < p.rps ? orow[nt] : p.rps - 1;
        rpre[nt] = *reinterpret_cast<const uint4*>(resn + rr * p.C_out + o0);
      }
    }
    float b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = 0.f;
    if (p.bias) {
      VecIO<float, 4>::load(p.bias + o0, reinterpret_cast<float(&)[4]>(b[0]));
      VecIO<float, 4>::load(p.bias + o0 + 4, reinterpret_cast<float(&)[4]>(b[4]));
    }
    f32x4_t acc[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[0][nt] = f32x4_t{b[0], b[1], b[2], b[3]};
      acc[1][nt] = f32x4_t{b[4], b[5], b[6], b[7]};
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t a = p.w[((long)(pr * 2 + mt) * KS + ks) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Mma<bf16_t>::mma(a, bact[ks][nt], acc[mt][nt]);
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (orow[nt] >= p.rps) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc[0][nt][j]; v[4 + j] = acc[1][nt][j]; }
      if (pre_ok) {
        float pre[8];
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rpre[nt]), pre);
        finish_and_store<bf16_t, 8, true>(v, p.e, n, orow[nt], o0, pre);
      } else {
        finish_and_store<bf16_t, 8, true>(v, p.e, n, orow[nt], o0);
      }
    }
  }
}

template <int KS, int NT>
static void launch_fast(const PwFastParams& p, int N, hipStream_t s) {
  const long rows_per_block = 4L * NT * 16;
  const long row_blocks = (p.rps + rows_per_block - 1) / rows_per_block * N;
  // deep levels (level 3: 172 row blocks, bottleneck: 22 for 256 CUs): the output-channel pairs are shared out over blockIdx.z
  // until ~2 workgroups per CU exist; every share re-reads the operand rows (L2 hits: the whole tensor is a few MB there)
  int zsplit = 1;
  if (tuning_get("pw_fast_zsplit", 1))
    while (row_blocks * zsplit < 512 && (p.C_out / 32) % (zsplit * 2) == 0) zsplit *= 2;
  dim3 grid((unsigned)((p.rps + rows_per_block - 1) / rows_per_block), (unsigned)N, (unsigned)zsplit), block(256);
  hipLaunchKernelGGL((pw_fast_kernel<KS, NT>), grid, block, 0, s, p);
}

bool pw_fast_supported(const pytc_pw_args* a) {
  if (a->in_dtype != PYTC_BF16 || a->out_dtype != PYTC_BF16 || a->w_dtype != PYTC_BF16) return false;
  if (a->C_in % 32 || a->C_out % 32 || (a->gather != 0 && a->gather != 2) || a->act != PYTC_ACT_NONE) return false;
  if (a->gather == 2 && a->res_mode == PYTC_RES_UPSAMPLE) return false;
  const int ks = a->C_in / 32;
  return ks == 1 || ks == 2 || ks == 4 || ks == 8 || ks == 16 || ks == 32;
}

void pw_fast_launch(const pytc_pw_args* a, const EpiParams& e, hipStream_t s) {
  PwFastParams p;
  p.x = (const bf16_t*)a->x; p.w = (const bf16x8_t*)a->w_packed; p.bias = a->bias; p.ab = a->ab; p.e = e;
  p.rps = a->rows_per_sample; p.rps_in = a->rows_per_sample; p.C_in = a->C_in; p.C_out = a->C_out; p.pre_act = a->pre_act;
  p.gather = a->gather; p.Hi = p.Wi = p.Ho = p.Wo = 0;
  if (a->gather == 2) {
    p.Hi = a->Hi; p.Wi = a->Wi; p.Ho = (a->Hi - 1) / 2 + 1; p.Wo = (a->Wi - 1) / 2 + 1;
    p.rps_in = (long)a->Di * a->Hi * a->Wi;
  }
  const int nt_knob = tuning_get("pw_fast_nt", 0);
  switch (a->C_in / 32) {
    case 1: if (nt_knob == 2) launch_fast<1, 2>(p, a->N, s); else launch_fast<1, 4>(p, a->N, s); break;
    case 2: if (nt_knob == 2) launch_fast<2, 2>(p, a->N, s); else launch_fast<2, 4>(p, a->N, s); break;
    case 4: if (nt_knob == 4) launch_fast<4, 4>(p, a->N, s); else launch_fast<4, 2>(p, a->N, s); break;
    case 8: launch_fast<8, 2>(p, a->N, s); break;
    case 16: launch_fast<16, 1>(p, a->N, s); break;
    default: launch_fast<32, 1>(p, a->N, s); break;
  }
}

}  // namespace pytc
