// Pointwise (1x1x1) convolutions as MFMA GEMMs in "transposed" form:
//     Y^T[o][v] = sum_k W[o][k] * X^T[k][v]
// i.e. the WEIGHTS are the MFMA A operand (M = output channels) and the NDHWC activations are the
// B operand (K = input channels, N = voxels).  Consequences on gfx950 (wave64):
//   * a lane's B fragment is EPL consecutive channels of ONE voxel = one 16-byte global load,
//     a wave's B-fragment load is fully coalesced, no LDS staging, no transposes;
//   * the accumulator (C/D layout: col = lane&15 -> voxel, row = 4*(lane>>4)+r -> channel) holds
//     4 consecutive output channels of one voxel per lane = one 8/16-byte NDHWC store.
// bf16 uses v_mfma_f32_16x16x32_bf16, fp32 uses 4x v_mfma_f32_16x16x4_f32 per 16-wide k-group
// (exact fp32, used for the tight parity gate).
#include "pw_common.h"

namespace pytc {

struct PwParams {
  const void* x;
  const void* wp;
  const float* bias;
  const float* ab;
  EpiParams e;
  long rps_out, rps_in;  // rows per sample
  int N, C_in, C_out, KG, MTt;
  int act, gather, pre_act;
  int Di, Hi, Wi;        // gather==2: input grid
  int Do, Ho, Wo;        // gather==2: output grid
};

template <typename TI, typename TW, typename TO, int MT, int NT>
__global__ void __launch_bounds__(256)
pw_conv_kernel(PwParams p) {
  typedef Mma<TW> M;
  constexpr int EPL = M::EPL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.z;
  const int mt0 = blockIdx.y * MT;
  const long row0 = ((long)blockIdx.x * 4 + wave) * (NT * 16);
  if (row0 >= p.rps_out) return;
  const int r = lane & 15, kb = lane >> 4;

  long orow[NT];
  const TI* xrow[NT];
  const TI* xn = reinterpret_cast<const TI*>(p.x) + (long)n * p.rps_in * p.C_in;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    long o = row0 + nt * 16 + r;
    orow[nt] = o;
    long oc = o < p.rps_out ? o : p.rps_out - 1;
    long src = oc;
    if (p.gather == 2) {
      int ox = (int)(oc % p.Wo);
      long t = oc / p.Wo;
      int oy = (int)(t % p.Ho);
      int oz = (int)(t / p.Ho);
      src = ((long)(2 * oz) * p.Hi + 2 * oy) * p.Wi + 2 * ox;
    }
    xrow[nt] = xn + src * p.C_in;
  }
  const bool vec_ok = (p.C_in % EPL) == 0;

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const typename M::frag_t* wp = reinterpret_cast<const typename M::frag_t*>(p.wp);
  for (int kg = 0; kg < p.KG; ++kg) {
    const int k0 = kg * M::KSTEP + kb * EPL;
    float av[EPL], bv[EPL];
    if (p.ab) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        bool ok = k0 + j < p.C_in;
        av[j] = ok ? p.ab[((long)n * 2 + 0) * p.C_in + k0 + j] : 0.f;
        bv[j] = ok ? p.ab[((long)n * 2 + 1) * p.C_in + k0 + j] : 0.f;
      }
    }
    typename M::frag_t bf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[EPL];
      load_row_frag<TI, EPL>(xrow[nt], k0, p.C_in, vec_ok, v);
      if (p.ab) {
#pragma unroll
        for (int j = 0; j < EPL; ++j) v[j] = fmaf(v[j], av[j], bv[j]);
      }
      if (p.pre_act == PYTC_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < EPL; ++j) v[j] = gelu_erf(v[j]);
      }
      bf[nt] = M::from_floats(v);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt0 + mt < p.MTt) {
        typename M::frag_t af = wp[((long)(mt0 + mt) * p.KG + kg) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = M::mma(af, bf[nt], acc[mt][nt]);
      }
    }
  }

  // ---- epilogue: lane holds channels o0..o0+3 of voxel orow[nt]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int o0 = (mt0 + mt) * 16 + kb * 4;
    if (o0 >= p.C_out) continue;
    float bo[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) bo[rr] = (p.bias && o0 + rr < p.C_out) ? p.bias[o0 + rr] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (orow[nt] >= p.rps_out) continue;
      float v[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) v[rr] = apply_act(acc[mt][nt][rr] + bo[rr], p.act);
      finish_and_store<TO, 4, true>(v, p.e, n, orow[nt], o0);
    }
  }
}

// ---- thin pointwise convs: C_in == 1 (stem) and C_out == 1 (one-channel heads) -------------------------------------
// No GEMM here, just a streaming kernel at 16 bytes per lane on the wide side; weights come from the packed MFMA image
// (element (o,k) = tile o/16, k-group k/KSTEP, lane (o%16) + 16*((k%KSTEP)/EPL), slot k%EPL), so callers pack as usual
// and the values are the same rounded weights the MFMA path would use.
template <typename TW>
__device__ __forceinline__ float packed_weight(const TW* wp, int KG, int o, int k) {
  typedef Mma<TW> M;
  const int kg = k / M::KSTEP, kr = k % M::KSTEP;
  return to_f32<TW>(wp[(((long)(o / 16) * KG + kg) * 64 + (o % 16) + 16 * (kr / M::EPL)) * M::EPL + kr % M::EPL]);
}

// y[r][o] = act(w[o] * T(x[r]) + b[o]);  lane = (row, chunk of 8 output channels); grid-stride over rows with the lane's
// 8 weights and biases in registers (256 % chunks == 0 keeps a lane on its chunk)
template <typename TI, typename TW, typename TO>
__global__ void __launch_bounds__(256)
pw_stem_kernel(const TI* __restrict__ x, const TW* __restrict__ wp, const float* __restrict__ bias, TO* __restrict__ y,
               long rows_total, int C_out, int act) {
  const int chunks = C_out / 8;
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  const int o0 = (int)(tid % chunks) * 8;
  const long rstride = (long)gridDim.x * 256 / chunks;
  float wv[8], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wv[j] = packed_weight<TW>(wp, 1, o0 + j, 0); bv[j] = bias ? bias[o0 + j] : 0.f; }
#pragma unroll 4
  for (long r = tid / chunks; r < rows_total; r += rstride) {
    // the MFMA path rounds the activation to the weight type before multiplying
    const float xv = to_f32<TW>(from_f32<TW>(to_f32<TI>(x[r])));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = apply_act(fmaf(wv[j], xv, bv[j]), act);
    if constexpr (sizeof(TO) == 2) VecIO<TO, 8>::store(y + r * C_out + o0, v);
    else {
      VecIO<float, 4>::store(reinterpret_cast<float*>(y) + r * C_out + o0, reinterpret_cast<float(&)[4]>(v[0]));
      VecIO<float, 4>::store(reinterpret_cast<float*>(y) + r * C_out + o0 + 4, reinterpret_cast<float(&)[4]>(v[4]));
    }
  }
}

// y[r] = act(sum_k w[k] * x[r][k] + b);  lane = (row, chunk of 8 input channels), xor-shuffle sum over the chunks
template <typename TW, typename TO>
__global__ void __launch_bounds__(256)
pw_head_kernel(const bf16_t* __restrict__ x, const TW* __restrict__ wp, const float* __restrict__ bias,
               TO* __restrict__ y, long rows_total, int C_in, int act) {
  const int chunks = C_in / 8;                      // power of two <= 64 (checked by the host)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long r = i / chunks;
  const int k0 = (int)(i % chunks) * 8;
  const int KG = (C_in + Mma<TW>::KSTEP - 1) / Mma<TW>::KSTEP;
  float s = 0.f;
  if (r < rows_total) {
    float v[8];
    VecIO<bf16_t, 8>::load(x + r * C_in + k0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(packed_weight<TW>(wp, KG, 0, k0 + j), v[j], s);
  }
  for (int off = 1; off < chunks; off <<= 1) s += __shfl_xor(s, off, 64);
  if (r < rows_total && k0 == 0) y[r] = from_f32<TO>(apply_act(s + (bias ? bias[0] : 0.f), act));
}

template <typename TW>
__global__ void __launch_bounds__(256)
pw_pack_kernel(const float* __restrict__ w, int C_out, int C_in, int transposed, TW* __restrict__ packed,
               int KG, long total) {
  typedef Mma<TW> M;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int j = (int)(i % M::EPL);
  long t = i / M::EPL;
  int lane = (int)(t % 64);
  t /= 64;
  int kg = (int)(t % KG);
  int mt = (int)(t / KG);
  int o = mt * 16 + (lane & 15);
  int k = kg * M::KSTEP + (lane >> 4) * M::EPL + j;
  float v = 0.f;
  if (o < C_out && k < C_in) v = transposed ? w[(long)k * C_out + o] : w[(long)o * C_in + k];
  packed[i] = from_f32<TW>(v);
}

template <typename TI, typename TW, typename TO, int NT>
static void launch_pw(const PwParams& p, int MT, hipStream_t s) {
  long rows_per_block = 4L * NT * 16;
  dim3 grid((unsigned)((p.rps_out + rows_per_block - 1) / rows_per_block), (unsigned)((p.MTt + MT - 1) / MT),
            (unsigned)p.N);
  dim3 block(256);
  switch (MT) {
    case 1: hipLaunchKernelGGL((pw_conv_kernel<TI, TW, TO, 1, NT>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((pw_conv_kernel<TI, TW, TO, 2, NT>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((pw_conv_kernel<TI, TW, TO, 4, NT>), grid, block, 0, s, p); break;
  }
}

bool pw_fast_supported(const pytc_pw_args* a);
void pw_fast_launch(const pytc_pw_args* a, const EpiParams& e, hipStream_t s);
bool pw_gemm_rowmajor_supported(const pytc_pw_args* a);          // pw_gemm_kernels.hip
void pw_gemm_rowmajor_launch(const pytc_pw_args* a, const EpiParams& e, hipStream_t s);

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_pw_conv_paired_supported(const pytc_pw_args* a) { return a && pw_fast_supported(a) ? 1 : 0; }
extern "C" int pytc_pw_conv_rowmajor_supported(const pytc_pw_args* a) { return a && pw_gemm_rowmajor_supported(a) ? 1 : 0; }

static int kstep_of(int dtype) { return dtype == PYTC_BF16 ? 32 : 16; }

extern "C" int64_t pytc_pw_packed_elems(int C_out, int C_in, int dtype) {
  if (C_out < 1 || C_in < 1 || (dtype != PYTC_F32 && dtype != PYTC_BF16)) return -1;
  int ks = kstep_of(dtype);
  return (int64_t)((C_out + 15) / 16) * 16 * ((C_in + ks - 1) / ks) * ks;
}

extern "C" int pytc_pw_pack_weight(const float* w, int C_out, int C_in, int transposed, void* packed, int dtype,
                                   void* stream) {
  PYTC_REQUIRE(w && packed && C_out >= 1 && C_in >= 1, "pw_pack_weight: bad arguments");
  PYTC_REQUIRE(dtype == PYTC_F32 || dtype == PYTC_BF16, "pw_pack_weight: bad dtype");
  long total = pytc_pw_packed_elems(C_out, C_in, dtype);
  int KG = (C_in + kstep_of(dtype) - 1) / kstep_of(dtype);
  dim3 grid(ceil_div(total, 256)), block(256);
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(pw_pack_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, transposed,
                       (bf16_t*)packed, KG, total);
  else
    hipLaunchKernelGGL(pw_pack_kernel<float>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, transposed,
                       (float*)packed, KG, total);
  PYTC_LAUNCH_CHECK("pw_pack_weight");
  return PYTC_OK;
}

extern "C" int pytc_pw_conv_fwd(const pytc_pw_args* a, void* stream) {
  PYTC_REQUIRE(a && a->x && a->w_packed && a->y, "pw_conv: null pointer");
  PYTC_REQUIRE(a->N >= 1 && a->rows_per_sample >= 1 && a->C_in >= 1 && a->C_out >= 1, "pw_conv: bad shape");
  PYTC_REQUIRE(a->w_dtype == PYTC_F32 || a->w_dtype == PYTC_BF16, "pw_conv: bad w_dtype");
  PYTC_REQUIRE(a->res_mode == PYTC_RES_NONE || a->res, "pw_conv: residual mode without residual pointer");
  PYTC_REQUIRE(a->res_mode != PYTC_RES_NORM_BWD || (a->w_paired && a->res_bias),
               "pw_conv: RES_NORM_BWD runs on the paired-row kernel and needs its coefficients in res_bias");
  PwParams p;
  p.x = a->x; p.wp = a->w_packed; p.bias = a->bias; p.ab = a->ab;
  p.e.res = a->res; p.e.res_low = a->res_low; p.e.res_bias = a->res_bias; p.e.y = a->y;
  p.e.rps_out = a->rows_per_sample; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode; p.e.nt = 0;
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  p.N = a->N; p.C_in = a->C_in; p.C_out = a->C_out;
  p.KG = (a->C_in + kstep_of(a->w_dtype) - 1) / kstep_of(a->w_dtype);
  p.MTt = (a->C_out + 15) / 16;
  p.act = a->act; p.gather = a->gather; p.pre_act = a->pre_act;
  PYTC_REQUIRE(a->pre_act == PYTC_ACT_NONE || a->pre_act == PYTC_ACT_GELU, "pw_conv: bad pre_act");
  p.rps_out = a->rows_per_sample; p.rps_in = a->rows_per_sample;
  p.Di = a->Di; p.Hi = a->Hi; p.Wi = a->Wi; p.Do = p.Ho = p.Wo = 0;
  if (a->gather == 2) {
    PYTC_REQUIRE(a->Di >= 1 && a->Hi >= 1 && a->Wi >= 1, "pw_conv: gather needs the input grid");
    p.Do = (a->Di - 1) / 2 + 1; p.Ho = (a->Hi - 1) / 2 + 1; p.Wo = (a->Wi - 1) / 2 + 1;
    PYTC_REQUIRE((long)p.Do * p.Ho * p.Wo == a->rows_per_sample, "pw_conv: gather grid does not match rows");
    p.rps_in = (long)a->Di * a->Hi * a->Wi;
  } else {
    PYTC_REQUIRE(a->gather == 0, "pw_conv: bad gather mode %d", a->gather);
  }
  if (a->res_mode == PYTC_RES_UPSAMPLE) {
    PYTC_REQUIRE(a->gather == 0, "pw_conv: RES_UPSAMPLE cannot be combined with gather");
    PYTC_REQUIRE((long)a->Di * a->Hi * a->Wi == a->rows_per_sample && !(a->Di & 1) && !(a->Hi & 1) && !(a->Wi & 1),
                 "pw_conv: RES_UPSAMPLE needs the (even) output grid");
    PYTC_REQUIRE(a->rows_per_sample < (1L << 31), "pw_conv: RES_UPSAMPLE positions are 32-bit (rows per sample < 2^31)");
    p.e.Go_d = a->Di; p.e.Go_h = a->Hi; p.e.Go_w = a->Wi;
    p.e.Gl_d = a->Di / 2; p.e.Gl_h = a->Hi / 2; p.e.Gl_w = a->Wi / 2;
  }
  if (a->res_mode == PYTC_RES_NORM_BWD && (a->Di | a->Hi | a->Wi) != 0) {
    // cropped form (up blocks): rows = the padded (Di, Hi, Wi) grid, y = the compact (Di - 1, Hi - 1, Wi - 1) grid (pw_common.h)
    PYTC_REQUIRE(a->gather == 0 && a->Di >= 2 && a->Hi >= 2 && a->Wi >= 2 && (long)a->Di * a->Hi * a->Wi == a->rows_per_sample &&
                 a->rows_per_sample < (1L << 31), "pw_conv: RES_NORM_BWD crop grid does not match the rows");
    p.e.Go_d = a->Di; p.e.Go_h = a->Hi; p.e.Go_w = a->Wi;
  }
  int MT = p.MTt >= 4 ? 4 : (p.MTt >= 2 ? 2 : 1);
  hipStream_t s = (hipStream_t)stream;
  const int ti = a->in_dtype, tw = a->w_dtype, to = a->out_dtype;
  const bool plain = !a->ab && a->pre_act == PYTC_ACT_NONE && a->res_mode == PYTC_RES_NONE && a->gather == 0 && !a->w_paired &&
                     tuning_get("pw_thin", 1) != 0;
  const long rows_total = (long)a->N * a->rows_per_sample;
  if (plain && a->C_in == 1 && a->C_out % 8 == 0 && 256 % (a->C_out / 8) == 0 && tw == PYTC_BF16 && (ti == PYTC_F32 || ti == PYTC_BF16) && to == PYTC_BF16) {
    const long work = rows_total * (a->C_out / 8);
    const int blocks = (int)(ceil_div(work, 256) < 8192 ? ceil_div(work, 256) : 8192);
    if (ti == PYTC_F32) hipLaunchKernelGGL((pw_stem_kernel<float, bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, s, (const float*)a->x, (const bf16_t*)a->w_packed, a->bias, (bf16_t*)a->y, rows_total, a->C_out, a->act);
    else hipLaunchKernelGGL((pw_stem_kernel<bf16_t, bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)a->x, (const bf16_t*)a->w_packed, a->bias, (bf16_t*)a->y, rows_total, a->C_out, a->act);
    PYTC_LAUNCH_CHECK("pw_conv");
    return PYTC_OK;
  }
  if (plain && a->C_out == 1 && ti == PYTC_BF16 && tw == PYTC_BF16 && a->C_in >= 8 && a->C_in <= 512 && (a->C_in & (a->C_in - 1)) == 0) {
    const long work = rows_total * (a->C_in / 8);
    if (to == PYTC_F32) hipLaunchKernelGGL((pw_head_kernel<bf16_t, float>), dim3(ceil_div(work, 256)), dim3(256), 0, s, (const bf16_t*)a->x, (const bf16_t*)a->w_packed, a->bias, (float*)a->y, rows_total, a->C_in, a->act);
    else hipLaunchKernelGGL((pw_head_kernel<bf16_t, bf16_t>), dim3(ceil_div(work, 256)), dim3(256), 0, s, (const bf16_t*)a->x, (const bf16_t*)a->w_packed, a->bias, (bf16_t*)a->y, rows_total, a->C_in, a->act);
    PYTC_LAUNCH_CHECK("pw_conv");
    return PYTC_OK;
  }
  if (a->w_paired == 2) {
    // plain row-major bf16 weight matrix [C_out][C_in]: the LDS-tiled GEMM (pw_gemm_kernels.hip), every prologue / epilogue of the paired-row kernel
    PYTC_REQUIRE(pw_gemm_rowmajor_supported(a), "pw_conv: w_paired = 2 (row-major weights, LDS-tiled GEMM) needs bf16, C_in %% 64 == 0, C_out %% 128 == 0, "
                 "no gather, no activation (got %d -> %d)", a->C_in, a->C_out);
    pw_gemm_rowmajor_launch(a, p.e, s);
    PYTC_LAUNCH_CHECK("pw_conv");
    return PYTC_OK;
  }
  if (a->w_paired) {
    PYTC_REQUIRE(pw_fast_supported(a), "pw_conv: w_paired set for a shape the paired-row kernel does not cover");
    pw_fast_launch(a, p.e, s);
    PYTC_LAUNCH_CHECK("pw_conv");
    return PYTC_OK;
  }
  if (ti == PYTC_F32 && tw == PYTC_F32 && to == PYTC_F32) launch_pw<float, float, float, 4>(p, MT, s);
  else if (ti == PYTC_BF16 && tw == PYTC_BF16 && to == PYTC_BF16) launch_pw<bf16_t, bf16_t, bf16_t, 4>(p, MT, s);
  else if (ti == PYTC_BF16 && tw == PYTC_BF16 && to == PYTC_F32) launch_pw<bf16_t, bf16_t, float, 4>(p, MT, s);
  else if (ti == PYTC_F32 && tw == PYTC_BF16 && to == PYTC_BF16) launch_pw<float, bf16_t, bf16_t, 4>(p, MT, s);
  else {
    set_error("pw_conv: unsupported dtype combination in=%d w=%d out=%d", ti, tw, to);
    return PYTC_ERR_UNSUPPORTED;
  }
  PYTC_LAUNCH_CHECK("pw_conv");
  return PYTC_OK;
}
