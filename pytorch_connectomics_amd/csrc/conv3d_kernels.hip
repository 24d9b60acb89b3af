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

template <typename TW>
__global__ void __launch_bounds__(256)
conv3d_pack_kernel(const float* __restrict__ w, int C_out, int C_in, int ntap, TW* __restrict__ packed, int KG,
                   long total) {
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
  if (o < C_out && k < C_in) v = w[((long)o * C_in + k) * ntap + tap];
  packed[i] = from_f32<TW>(v);
}

template <typename TI, typename TW, typename TO>
static void launch_conv(const ConvParams& p, hipStream_t s) {
  constexpr int NT = 4;
  const long rps = (long)p.D * p.H * p.W;
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
  return (int64_t)((C_out + 15) / 16) * 16 * kd * kh * kw * ((C_in + ks - 1) / ks) * ks;
}

extern "C" int pytc_conv3d_pack_weight(const float* w, int C_out, int C_in, int kd, int kh, int kw, void* packed,
                                       int dtype, void* stream) {
  PYTC_REQUIRE(w && packed, "conv3d_pack_weight: null pointer");
  long total = pytc_conv3d_packed_elems(C_out, C_in, kd, kh, kw, dtype);
  PYTC_REQUIRE(total > 0, "conv3d_pack_weight: bad arguments");
  int KG = (C_in + kstep_of(dtype) - 1) / kstep_of(dtype);
  dim3 grid(ceil_div(total, 256)), block(256);
  if (dtype == PYTC_BF16)
    hipLaunchKernelGGL(conv3d_pack_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, kd * kh * kw,
                       (bf16_t*)packed, KG, total);
  else
    hipLaunchKernelGGL(conv3d_pack_kernel<float>, grid, block, 0, (hipStream_t)stream, w, C_out, C_in, kd * kh * kw,
                       (float*)packed, KG, total);
  PYTC_LAUNCH_CHECK("conv3d_pack_weight");
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
  p.e.res = a->res; p.e.res_low = nullptr; p.e.res_bias = nullptr; p.e.y = a->y;
  p.e.rps_out = (long)a->D * a->H * a->W; p.e.C_out = a->C_out; p.e.res_mode = a->res_mode;
  p.e.Go_d = p.e.Go_h = p.e.Go_w = p.e.Gl_d = p.e.Gl_h = p.e.Gl_w = 0;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == PYTC_F32) launch_conv<float, float, float>(p, s);
  else launch_conv<bf16_t, bf16_t, bf16_t>(p, s);
  PYTC_LAUNCH_CHECK("conv3d");
  return PYTC_OK;
}
