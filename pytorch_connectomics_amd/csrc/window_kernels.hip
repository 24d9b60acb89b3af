// Sliding-window kernels: window gather (+pad, +TTA view), overlap-add blend, finalize,
// TTA ensemble update.  All HBM-bound elementwise/scatter work: x-fastest thread mapping so
// that every global access is coalesced; no atomics (one launch per window, stream-ordered,
// so the fp32 accumulation order is exactly the reference's window order).
#include "pytc_common.h"

namespace pytc {

struct StartList {
  int s[64 * 3];
};

// Source index for global coordinate g on an axis whose in-volume part of THIS window is [lo, hi).
// The reference pads the window's inner crop (F.pad in window.py:464-527, np.pad in lazy.py:852-904), so
// reflect / replicate / circular are evaluated relative to that crop (periodic, like np.pad), not to the
// whole volume -- the two only differ when the overhang reaches the crop's far end.
__device__ __forceinline__ int pad_index(int g, int lo, int hi, int mode, bool& inside) {
  inside = true;
  if (g >= lo && g < hi) return g;
  const int n = hi - lo, j = g - lo;
  if (n <= 0) { inside = false; return 0; }
  switch (mode) {
    case PYTC_PAD_REFLECT: {
      if (n == 1) return lo;
      int p = 2 * (n - 1);
      int m = j % p;
      if (m < 0) m += p;
      return lo + (m < n ? m : p - m);
    }
    case PYTC_PAD_REPLICATE:
      return j < 0 ? lo : hi - 1;
    case PYTC_PAD_CIRCULAR: {
      int m = j % n;
      return lo + (m < 0 ? m + n : m);
    }
    default:
      inside = false;
      return 0;
  }
}

// view: out[z,y,x] = win[T(F(z,y,x))], F = per-axis flips, T = optional exchange of two axes ((y,x), (z,y) or (z,x): at most one)
__device__ __forceinline__ void view_src(int view, int rz, int ry, int rx, int z, int y, int x,
                                         int& wz, int& wy, int& wx) {
  int fz = (view & PYTC_VIEW_FLIP_Z) ? rz - 1 - z : z;
  int fy = (view & PYTC_VIEW_FLIP_Y) ? ry - 1 - y : y;
  int fx = (view & PYTC_VIEW_FLIP_X) ? rx - 1 - x : x;
  if (view & PYTC_VIEW_SWAP_YX) {
    wz = fz; wy = fx; wx = fy;
  } else if (view & PYTC_VIEW_SWAP_ZY) {
    wz = fy; wy = fz; wx = fx;
  } else if (view & PYTC_VIEW_SWAP_ZX) {
    wz = fx; wy = fy; wx = fz;
  } else {
    wz = fz; wy = fy; wx = fx;
  }
}

static inline bool view_ok(int view, int rz, int ry, int rx) {
  const int swaps = view & (PYTC_VIEW_SWAP_YX | PYTC_VIEW_SWAP_ZY | PYTC_VIEW_SWAP_ZX);
  if (view & ~63) return false;
  if (swaps == 0) return true;
  if (swaps == PYTC_VIEW_SWAP_YX) return ry == rx;
  if (swaps == PYTC_VIEW_SWAP_ZY) return rz == ry;
  if (swaps == PYTC_VIEW_SWAP_ZX) return rz == rx;
  return false;
}

template <typename TO>
__global__ void __launch_bounds__(256)
gather_windows_kernel(const float* __restrict__ vol, int C, int Z, int Y, int X, StartList st,
                      int rz, int ry, int rx, int view, int pad_mode, float cval, int pz, int py,
                      int px, TO* __restrict__ out) {
  // pz/py/px: per-axis "mode valid" flags (reflect/circular fall back to constant when the pad
  // would reach the inner extent, window.py:511-518) are resolved on the host per window batch:
  // here pad_mode already is the effective mode.
  const int b = blockIdx.z;
  const long per_win = (long)rz * ry * rx;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_win) return;
  int x = (int)(i % rx);
  long t = i / rx;
  int y = (int)(t % ry);
  int z = (int)(t / ry);
  int wz, wy, wx;
  view_src(view, rz, ry, rx, z, y, x, wz, wy, wx);
  const int s0 = st.s[3 * b + 0], s1 = st.s[3 * b + 1], s2 = st.s[3 * b + 2];
  int gz = s0 + wz, gy = s1 + wy, gx = s2 + wx;
  const int lz = max(0, s0), hz = min(Z, s0 + rz), ly = max(0, s1), hy = min(Y, s1 + ry);
  const int lx = max(0, s2), hx = min(X, s2 + rx);
  // np.pad 'reflect' on a crop with a length-1 axis degrades to 'edge' for the whole pad (lazy.py:248-253)
  if (pad_mode == PYTC_PAD_REFLECT && (hz - lz <= 1 || hy - ly <= 1 || hx - lx <= 1)) pad_mode = PYTC_PAD_REPLICATE;
  bool iz, iy, ix;
  int sz = pad_index(gz, lz, hz, pad_mode, iz);
  int sy = pad_index(gy, ly, hy, pad_mode, iy);
  int sx = pad_index(gx, lx, hx, pad_mode, ix);
  bool inside = iz && iy && ix;
  TO* o = out + ((long)b * per_win + i) * C;
  const long plane = (long)Z * Y * X;
  long src = ((long)sz * Y + sy) * X + sx;
  for (int c = 0; c < C; ++c) {
    float v = inside ? vol[c * plane + src] : cval;
    o[c] = from_f32<TO>(v);
  }
}

template <typename TP>
__global__ void __launch_bounds__(256)
blend_accumulate_kernel(const TP* __restrict__ pred, int sz, int sy, int sx, int rz, int ry, int rx,
                        int C, int view, const float* __restrict__ wzv, const float* __restrict__ wyv,
                        const float* __restrict__ wxv, int combine, float floor_w, int bz, int by, int bx,
                        float* __restrict__ value, float* __restrict__ weight, int Z, int Y, int X) {
  const long per_win = (long)rz * ry * rx;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_win) return;
  int x = (int)(i % rx);
  long t = i / rx;
  int y = (int)(t % ry);
  int z = (int)(t / ry);
  int wz, wy, wx;
  view_src(view, rz, ry, rx, z, y, x, wz, wy, wx);
  int gz = sz + wz, gy = sy + wy, gx = sx + wx;
  if (gz < 0 || gz >= Z || gy < 0 || gy >= Y || gx < 0 || gx >= X) return;
  float w;
  if (combine == PYTC_BLEND_MIN) {
    w = fminf(fminf(wzv[wz], wyv[wy]), wxv[wx]);
  } else {
    // (wz*wy)*wx, each product rounded (axis order of window.py:178-194), then the two floors
    w = __fmul_rn(__fmul_rn(wzv[wz], wyv[wy]), wxv[wx]);
    w = fmaxf(w, 1.17549435e-38f);
    w = fmaxf(w, floor_w);
  }
  // border mask: the outer b voxels of the window map are exactly zero (window.py:297-319, applied after
  // the floors in the lazy path)
  if (wz < bz || wz >= rz - bz || wy < by || wy >= ry - by || wx < bx || wx >= rx - bx) w = 0.f;
  const long plane = (long)Z * Y * X;
  long dst = ((long)gz * Y + gy) * X + gx;
  const TP* p = pred + i * C;
  for (int c = 0; c < C; ++c) {
    float pv = to_f32<TP>(p[c]);
    // separate multiply and add roundings: value += pred * w  (window.py:652-654)
    value[c * plane + dst] = __fadd_rn(value[c * plane + dst], __fmul_rn(pv, w));
  }
  if (weight) weight[dst] = __fadd_rn(weight[dst], w);
}

// ---- affinity-aware blending (inference/tta_affinity.py:350-393 fused into the scatter) ------------------------------
// Output channel d of the canonical window takes prediction channel src[d], displaced by shift[d]: the value predicted
// at canonical position q lands at p = q + shift[d] and is weighted by the blending map AT p (the reference multiplies
// the re-anchored patch by the map); positions with p outside the window do not exist (the wrapped face).
constexpr int MAX_MAP = 32;
struct ChanMap { int src[MAX_MAP]; int sz[MAX_MAP]; int sy[MAX_MAP]; int sx[MAX_MAP]; };

__device__ __forceinline__ float window_weight(const float* wzv, const float* wyv, const float* wxv, int combine,
                                               float floor_w, int z, int y, int x, int rz, int ry, int rx, int bz,
                                               int by, int bx) {
  float w;
  if (combine == PYTC_BLEND_MIN) {
    w = fminf(fminf(wzv[z], wyv[y]), wxv[x]);
  } else {
    w = __fmul_rn(__fmul_rn(wzv[z], wyv[y]), wxv[x]);
    w = fmaxf(w, 1.17549435e-38f);
    w = fmaxf(w, floor_w);
  }
  if (z < bz || z >= rz - bz || y < by || y >= ry - by || x < bx || x >= rx - bx) w = 0.f;
  return w;
}

template <typename TP>
__global__ void __launch_bounds__(256)
blend_accumulate_mapped_kernel(const TP* __restrict__ pred, int sz, int sy, int sx, int rz, int ry, int rx, int C,
                               int view, const float* __restrict__ wzv, const float* __restrict__ wyv,
                               const float* __restrict__ wxv, int combine, float floor_w, int bz, int by, int bx,
                               ChanMap m, float* __restrict__ value, float* __restrict__ weight, int Z, int Y, int X) {
  const long per_win = (long)rz * ry * rx;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_win) return;
  const int x = (int)(i % rx);
  const long t = i / rx;
  const int y = (int)(t % ry), z = (int)(t / ry);
  int qz, qy, qx;
  view_src(view, rz, ry, rx, z, y, x, qz, qy, qx);
  const long plane = (long)Z * Y * X;
  const TP* p = pred + i * C;
  for (int d = 0; d < C; ++d) {
    const int pz = qz + m.sz[d], py = qy + m.sy[d], px = qx + m.sx[d];
    if (pz < 0 || pz >= rz || py < 0 || py >= ry || px < 0 || px >= rx) continue;
    const int gz = sz + pz, gy = sy + py, gx = sx + px;
    if (gz < 0 || gz >= Z || gy < 0 || gy >= Y || gx < 0 || gx >= X) continue;
    const float w = window_weight(wzv, wyv, wxv, combine, floor_w, pz, py, px, rz, ry, rx, bz, by, bx);
    const long dst = ((long)gz * Y + gy) * X + gx;
    value[d * plane + dst] = __fadd_rn(value[d * plane + dst], __fmul_rn(to_f32<TP>(p[m.src[d]]), w));
  }
  if (weight) {
    const int gz = sz + qz, gy = sy + qy, gx = sx + qx;
    if (gz >= 0 && gz < Z && gy >= 0 && gy < Y && gx >= 0 && gx < X) {
      const long dst = ((long)gz * Y + gy) * X + gx;
      weight[dst] = __fadd_rn(weight[dst], window_weight(wzv, wyv, wxv, combine, floor_w, qz, qy, qx, rz, ry, rx, bz, by, bx));
    }
  }
}

// weight[g(p)] += w(p) for the positions p of a window whose source p - shift lies inside the window
__global__ void __launch_bounds__(256)
blend_weight_shifted_kernel(int sz, int sy, int sx, int rz, int ry, int rx, const float* __restrict__ wzv,
                            const float* __restrict__ wyv, const float* __restrict__ wxv, int combine, float floor_w,
                            int bz, int by, int bx, int hz, int hy, int hx, float* __restrict__ weight, int Z, int Y, int X) {
  const long per_win = (long)rz * ry * rx;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_win) return;
  const int px = (int)(i % rx);
  const long t = i / rx;
  const int py = (int)(t % ry), pz = (int)(t / ry);
  const int qz = pz - hz, qy = py - hy, qx = px - hx;
  if (qz < 0 || qz >= rz || qy < 0 || qy >= ry || qx < 0 || qx >= rx) return;
  const int gz = sz + pz, gy = sy + py, gx = sx + px;
  if (gz < 0 || gz >= Z || gy < 0 || gy >= Y || gx < 0 || gx >= X) return;
  const long dst = ((long)gz * Y + gy) * X + gx;
  weight[dst] = __fadd_rn(weight[dst], window_weight(wzv, wyv, wxv, combine, floor_w, pz, py, px, rz, ry, rx, bz, by, bx));
}

// v = w > 0 ? v / w : 0   (tta.py:1238-1244: partial channels are normalised by their own coverage, unclamped)
__global__ void __launch_bounds__(256)
normalize_covered_kernel(float* __restrict__ value, const float* __restrict__ weight, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) value[i] = weight[i] > 0.f ? __fdiv_rn(value[i], weight[i]) : 0.f;
}

// validity-aware running statistics (tta_ensemble.py:122-165): where cover > 0: mean -> stat += x, min/max, count += 1
__global__ void __launch_bounds__(256)
ensemble_update_masked_kernel(float* __restrict__ stat, float* __restrict__ count, const float* __restrict__ x,
                              const float* __restrict__ cover, long n, int mode) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    if (cover && !(cover[i] > 0.f)) continue;
    const float v = x[i];
    float a = stat[i];
    a = mode == 0 ? __fadd_rn(a, v) : (mode == 1 ? fminf(a, v) : fmaxf(a, v));
    stat[i] = a;
    count[i] = count[i] + 1.f;
  }
}

// mean: out = stat / count;  min/max: out = stat   (tta_ensemble.py:205-210; count == 0 is checked by the caller)
__global__ void __launch_bounds__(256)
ensemble_finalize_masked_kernel(const float* __restrict__ stat, const float* __restrict__ count, float* __restrict__ out,
                                long n, int mode) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = mode == 0 ? __fdiv_rn(stat[i], count[i]) : stat[i];
}

__global__ void __launch_bounds__(256)
blend_finalize_kernel(float* __restrict__ value, const float* __restrict__ weight, int C, long nvox,
                      float clamp, int act) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < nvox; i += stride) {
    float d = fmaxf(weight[i], clamp);
    for (int c = 0; c < C; ++c) {
      float v = __fdiv_rn(value[c * nvox + i], d);
      if (act == PYTC_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
      else if (act == PYTC_ACT_TANH) v = tanhf(v);
      value[c * nvox + i] = v;
    }
  }
}

__global__ void __launch_bounds__(256)
ensemble_update_kernel(float* __restrict__ acc, const float* __restrict__ x, long n, int mode, int count) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float a = acc[i], v = x[i];
    if (count <= 1) a = v;
    else if (mode == 0) a = __fadd_rn(a, __fdiv_rn(__fsub_rn(v, a), (float)count));
    else if (mode == 1) a = fminf(a, v);
    else a = fmaxf(a, v);
    acc[i] = a;
  }
}

// value [C][nvox]; channels [c0, c1): act in place.  softmax runs across the channel group per voxel.
__global__ void __launch_bounds__(256)
channel_activation_kernel(float* __restrict__ value, long nvox, long cs, long vs, int c0, int c1, int act,
                          float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < nvox; i += stride) {
    if (act == 4) {   // softmax over [c0, c1)
      float m = -3.402823466e38f;
      for (int c = c0; c < c1; ++c) m = fmaxf(m, value[c * cs + i * vs]);
      float s = 0.f;
      for (int c = c0; c < c1; ++c) s += expf(value[c * cs + i * vs] - m);
      for (int c = c0; c < c1; ++c) value[c * cs + i * vs] = expf(value[c * cs + i * vs] - m) / s;
    } else {
      for (int c = c0; c < c1; ++c) {
        float v = value[c * cs + i * vs] * scale;
        if (act == PYTC_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        else if (act == PYTC_ACT_TANH) v = tanhf(v);
        value[c * cs + i * vs] = v;
      }
    }
  }
}

// channels-last value [nvox][C], elementwise activations (no softmax): the tensor as a flat float4 stream -- a thread per voxel walking its
// C channels touched 28-byte pieces 28 bytes apart (7 channels: 1.5 TB/s, 308 us per 2 x 160^3 x 7 window batch of the lazy loop); same
// expressions per element
__global__ void __launch_bounds__(256)
channel_activation_flat_kernel(float4* __restrict__ value, long n4, int C, int c0, int c1, int act, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 q = value[i];
    float v[4] = {q.x, q.y, q.z, q.w};
    int c = (int)((i * 4) % C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c >= c0 && c < c1) {
        float t = v[j] * scale;
        if (act == PYTC_ACT_SIGMOID) t = 1.0f / (1.0f + expf(-t));
        else if (act == PYTC_ACT_TANH) t = tanhf(t);
        v[j] = t;
      }
      if (++c == C) c = 0;
    }
    value[i] = float4{v[0], v[1], v[2], v[3]};
  }
}

}  // namespace pytc

using namespace pytc;

extern "C" int pytc_channel_activation(float* value, int C, int64_t nvox, int channels_last, int c0, int c1, int act,
                                       float scale, void* stream) {
  PYTC_REQUIRE(value && C >= 1 && nvox > 0 && c0 >= 0 && c1 <= C && c0 < c1, "channel_activation: bad arguments");
  PYTC_REQUIRE(act == PYTC_ACT_NONE || act == PYTC_ACT_SIGMOID || act == PYTC_ACT_TANH || act == 4,
               "channel_activation: bad activation %d", act);
  if (channels_last && act != 4 && ((long)nvox * C) % 4 == 0 && ((uintptr_t)value & 15) == 0 && tuning_get("channel_act_flat", 1)) {
    const long n4 = (long)nvox * C / 4;
    const long want = (n4 + 255) / 256;
    hipLaunchKernelGGL(channel_activation_flat_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(value), n4, C, c0, c1, act, scale);
    PYTC_LAUNCH_CHECK("channel_activation");
    return PYTC_OK;
  }
  int blocks = (int)((nvox + 255) / 256 < 8192 ? (nvox + 255) / 256 : 8192);
  hipLaunchKernelGGL(channel_activation_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, value, (long)nvox,
                     channels_last ? 1L : (long)nvox, channels_last ? (long)C : 1L, c0, c1, act, scale);
  PYTC_LAUNCH_CHECK("channel_activation");
  return PYTC_OK;
}

extern "C" int pytc_gather_windows(const float* vol, int C, int Z, int Y, int X, const int32_t* starts,
                                   int B, int rz, int ry, int rx, int view, int pad_mode, float cval,
                                   void* out, int out_dtype, void* stream) {
  PYTC_REQUIRE(vol && out && starts, "gather_windows: null pointer");
  PYTC_REQUIRE(B >= 1 && B <= 64, "gather_windows: B=%d must be in [1,64]", B);
  PYTC_REQUIRE(C >= 1 && rz > 0 && ry > 0 && rx > 0 && Z > 0 && Y > 0 && X > 0, "gather_windows: bad shape");
  PYTC_REQUIRE(view_ok(view, rz, ry, rx), "gather_windows: at most one SWAP bit, and the exchanged window axes must have equal length");
  PYTC_REQUIRE(pad_mode >= 0 && pad_mode <= 3, "gather_windows: bad pad_mode %d", pad_mode);
  StartList st;
  memcpy(st.s, starts, sizeof(int) * 3 * B);
  long per_win = (long)rz * ry * rx;
  dim3 grid(ceil_div(per_win, 256), 1, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == PYTC_F32)
    hipLaunchKernelGGL(gather_windows_kernel<float>, grid, block, 0, s, vol, C, Z, Y, X, st, rz, ry, rx, view,
                       pad_mode, cval, 0, 0, 0, (float*)out);
  else if (out_dtype == PYTC_BF16)
    hipLaunchKernelGGL(gather_windows_kernel<bf16_t>, grid, block, 0, s, vol, C, Z, Y, X, st, rz, ry, rx, view,
                       pad_mode, cval, 0, 0, 0, (bf16_t*)out);
  else
    PYTC_REQUIRE(false, "gather_windows: bad out_dtype %d", out_dtype);
  PYTC_LAUNCH_CHECK("gather_windows");
  return PYTC_OK;
}

extern "C" int pytc_blend_accumulate(const void* pred, int pred_dtype, int B, const int32_t* starts, int rz,
                                     int ry, int rx, int C, int view, const float* wz, const float* wy,
                                     const float* wx, int combine, float floor_w, const int32_t* border,
                                     float* value, float* weight, int Z, int Y, int X, void* stream) {
  PYTC_REQUIRE(pred && starts && wz && wy && wx && value, "blend_accumulate: null pointer");
  PYTC_REQUIRE(B >= 1 && C >= 1, "blend_accumulate: bad B/C");
  PYTC_REQUIRE(view_ok(view, rz, ry, rx), "blend_accumulate: at most one SWAP bit, and the exchanged window axes must have equal length");
  PYTC_REQUIRE(combine == PYTC_BLEND_PRODUCT || combine == PYTC_BLEND_MIN, "blend_accumulate: bad combine");
  long per_win = (long)rz * ry * rx;
  dim3 grid(ceil_div(per_win, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int bz = border ? border[0] : 0, by = border ? border[1] : 0, bx = border ? border[2] : 0;
  PYTC_REQUIRE(bz >= 0 && by >= 0 && bx >= 0 && 2 * bz < rz && 2 * by < ry && 2 * bx < rx,
               "blend_accumulate: border mask too large for the window");
  for (int b = 0; b < B; ++b) {
    int sz = starts[3 * b], sy = starts[3 * b + 1], sx = starts[3 * b + 2];
    if (pred_dtype == PYTC_F32)
      hipLaunchKernelGGL(blend_accumulate_kernel<float>, grid, block, 0, s,
                         (const float*)pred + (long)b * per_win * C, sz, sy, sx, rz, ry, rx, C, view, wz, wy, wx,
                         combine, floor_w, bz, by, bx, value, weight, Z, Y, X);
    else if (pred_dtype == PYTC_BF16)
      hipLaunchKernelGGL(blend_accumulate_kernel<bf16_t>, grid, block, 0, s,
                         (const bf16_t*)pred + (long)b * per_win * C, sz, sy, sx, rz, ry, rx, C, view, wz, wy, wx,
                         combine, floor_w, bz, by, bx, value, weight, Z, Y, X);
    else
      PYTC_REQUIRE(false, "blend_accumulate: bad pred_dtype %d", pred_dtype);
  }
  PYTC_LAUNCH_CHECK("blend_accumulate");
  return PYTC_OK;
}

extern "C" int pytc_blend_accumulate_mapped(const void* pred, int pred_dtype, int B, const int32_t* starts, int rz, int ry,
                                            int rx, int C, int view, const float* wz, const float* wy, const float* wx,
                                            int combine, float floor_w, const int32_t* border, const int32_t* chan_src,
                                            const int32_t* chan_shift, float* value, float* weight, int Z, int Y, int X,
                                            void* stream) {
  PYTC_REQUIRE(pred && starts && wz && wy && wx && value && chan_src && chan_shift, "blend_accumulate_mapped: null pointer");
  PYTC_REQUIRE(B >= 1 && C >= 1 && C <= MAX_MAP, "blend_accumulate_mapped: C=%d must be in [1,%d]", C, MAX_MAP);
  PYTC_REQUIRE(view_ok(view, rz, ry, rx), "blend_accumulate_mapped: at most one SWAP bit, and the exchanged window axes must have equal length");
  PYTC_REQUIRE(combine == PYTC_BLEND_PRODUCT || combine == PYTC_BLEND_MIN, "blend_accumulate_mapped: bad combine");
  ChanMap m;
  for (int d = 0; d < C; ++d) {
    PYTC_REQUIRE(chan_src[d] >= 0 && chan_src[d] < C, "blend_accumulate_mapped: bad source channel %d", chan_src[d]);
    m.src[d] = chan_src[d]; m.sz[d] = chan_shift[3 * d]; m.sy[d] = chan_shift[3 * d + 1]; m.sx[d] = chan_shift[3 * d + 2];
  }
  const long per_win = (long)rz * ry * rx;
  dim3 grid(ceil_div(per_win, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int bz = border ? border[0] : 0, by = border ? border[1] : 0, bx = border ? border[2] : 0;
  PYTC_REQUIRE(bz >= 0 && by >= 0 && bx >= 0 && 2 * bz < rz && 2 * by < ry && 2 * bx < rx,
               "blend_accumulate_mapped: border mask too large for the window");
  for (int b = 0; b < B; ++b) {
    const int sz = starts[3 * b], sy = starts[3 * b + 1], sx = starts[3 * b + 2];
    if (pred_dtype == PYTC_F32)
      hipLaunchKernelGGL(blend_accumulate_mapped_kernel<float>, grid, block, 0, s, (const float*)pred + (long)b * per_win * C,
                         sz, sy, sx, rz, ry, rx, C, view, wz, wy, wx, combine, floor_w, bz, by, bx, m, value, weight, Z, Y, X);
    else if (pred_dtype == PYTC_BF16)
      hipLaunchKernelGGL(blend_accumulate_mapped_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)pred + (long)b * per_win * C,
                         sz, sy, sx, rz, ry, rx, C, view, wz, wy, wx, combine, floor_w, bz, by, bx, m, value, weight, Z, Y, X);
    else
      PYTC_REQUIRE(false, "blend_accumulate_mapped: bad pred_dtype %d", pred_dtype);
  }
  PYTC_LAUNCH_CHECK("blend_accumulate_mapped");
  return PYTC_OK;
}

extern "C" int pytc_blend_weight_shifted(int B, const int32_t* starts, int rz, int ry, int rx, const float* wz,
                                         const float* wy, const float* wx, int combine, float floor_w,
                                         const int32_t* border, const int32_t* shift, float* weight, int Z, int Y, int X,
                                         void* stream) {
  PYTC_REQUIRE(starts && wz && wy && wx && shift && weight && B >= 1, "blend_weight_shifted: bad arguments");
  const long per_win = (long)rz * ry * rx;
  const int bz = border ? border[0] : 0, by = border ? border[1] : 0, bx = border ? border[2] : 0;
  for (int b = 0; b < B; ++b)
    hipLaunchKernelGGL(blend_weight_shifted_kernel, dim3(ceil_div(per_win, 256)), dim3(256), 0, (hipStream_t)stream,
                       starts[3 * b], starts[3 * b + 1], starts[3 * b + 2], rz, ry, rx, wz, wy, wx, combine, floor_w, bz, by, bx,
                       shift[0], shift[1], shift[2], weight, Z, Y, X);
  PYTC_LAUNCH_CHECK("blend_weight_shifted");
  return PYTC_OK;
}

extern "C" int pytc_normalize_covered(float* value, const float* weight, int64_t n, void* stream) {
  PYTC_REQUIRE(value && weight && n > 0, "normalize_covered: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(normalize_covered_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, value, weight, (long)n);
  PYTC_LAUNCH_CHECK("normalize_covered");
  return PYTC_OK;
}

extern "C" int pytc_ensemble_update_masked(float* stat, float* count, const float* x, const float* cover, int64_t n,
                                           int mode, void* stream) {
  PYTC_REQUIRE(stat && count && x && n > 0 && mode >= 0 && mode <= 2, "ensemble_update_masked: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(ensemble_update_masked_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, stat, count, x, cover,
                     (long)n, mode);
  PYTC_LAUNCH_CHECK("ensemble_update_masked");
  return PYTC_OK;
}

extern "C" int pytc_ensemble_finalize_masked(const float* stat, const float* count, float* out, int64_t n, int mode,
                                             void* stream) {
  PYTC_REQUIRE(stat && count && out && n > 0 && mode >= 0 && mode <= 2, "ensemble_finalize_masked: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(ensemble_finalize_masked_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, stat, count, out,
                     (long)n, mode);
  PYTC_LAUNCH_CHECK("ensemble_finalize_masked");
  return PYTC_OK;
}

extern "C" int pytc_blend_finalize(float* value, const float* weight, int C, int64_t nvox, float clamp, int act,
                                   void* stream) {
  PYTC_REQUIRE(value && weight && C >= 1 && nvox > 0, "blend_finalize: bad arguments");
  int blocks = (int)((nvox + 255) / 256 < 8192 ? (nvox + 255) / 256 : 8192);
  hipLaunchKernelGGL(blend_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, value, weight, C,
                     (long)nvox, clamp, act);
  PYTC_LAUNCH_CHECK("blend_finalize");
  return PYTC_OK;
}

namespace pytc {
template <typename TO, bool INTEGER>
__global__ void __launch_bounds__(256)
scale_cast_kernel(const float* __restrict__ x, TO* __restrict__ y, long n, float scale, float lo, float hi) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i < n; i += stride) {
    float v[4];
    if (i + 4 <= n) {
      const f32x4_t t = *reinterpret_cast<const f32x4_t*>(x + i);
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = i + j < n ? x[i + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float u = __fmul_rn(v[j], scale);
      if (INTEGER) u = fminf(fmaxf(u, lo), hi);       // np.clip, then astype truncates toward zero
      if (i + j < n) y[i + j] = (TO)u;
    }
  }
}
}  // namespace pytc

extern "C" int pytc_scale_cast(const float* x, void* y, int64_t n, float scale, int target, void* stream) {
  PYTC_REQUIRE(x && y && n > 0, "scale_cast: bad arguments");
  const float sc = scale > 0.f ? scale : 1.f;
  const int blocks = (int)((n / 4 + 255) / 256 < 8192 ? (n / 4 + 255) / 256 + 1 : 8192);
  hipStream_t s = (hipStream_t)stream;
#define PYTC_SC(TT, INTG, LO, HI) hipLaunchKernelGGL((pytc::scale_cast_kernel<TT, INTG>), dim3(blocks), dim3(256), 0, s, x, (TT*)y, (long)n, sc, LO, HI)
  switch (target) {
    case PYTC_ST_U8: PYTC_SC(unsigned char, true, 0.f, 255.f); break;
    case PYTC_ST_I8: PYTC_SC(signed char, true, -128.f, 127.f); break;
    case PYTC_ST_U16: PYTC_SC(unsigned short, true, 0.f, 65535.f); break;
    case PYTC_ST_I16: PYTC_SC(short, true, -32768.f, 32767.f); break;
    case PYTC_ST_I32: PYTC_SC(int, true, -2147483648.f, 2147483520.f); break;   // largest fp32 below 2^31
    case PYTC_ST_F16: PYTC_SC(_Float16, false, 0.f, 0.f); break;
    case PYTC_ST_F32: PYTC_SC(float, false, 0.f, 0.f); break;
    default: set_error("scale_cast: unknown target %d", target); return PYTC_ERR_INVALID;
  }
#undef PYTC_SC
  PYTC_LAUNCH_CHECK("scale_cast");
  return PYTC_OK;
}

extern "C" int pytc_ensemble_update(float* acc, const float* x, int64_t n, int mode, int count, void* stream) {
  PYTC_REQUIRE(acc && x && n > 0 && mode >= 0 && mode <= 2 && count >= 1, "ensemble_update: bad arguments");
  int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(ensemble_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, x, (long)n, mode,
                     count);
  PYTC_LAUNCH_CHECK("ensemble_update");
  return PYTC_OK;
}
