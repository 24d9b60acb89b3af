// Train-step epilogue on the device:
//   * fused weighted-BCE-with-logits + sigmoid-Dice loss (models/losses/losses.py:190-266 + MONAI DiceLoss(sigmoid=True)
//     semantics, profiles/loss_profiles.yaml:2-9): ONE pass over (logits, target, mask) produces the five sums per
//     (sample, channel) both terms need, a one-workgroup finalize forms the scalar loss, ONE elementwise pass forms
//     dL/dlogits of both terms from the saved sums.  Strided operands (the network output is a channels-last tensor
//     viewed as NCDHW, labels are NCDHW).  Two-stage deterministic reductions, no float atomics.
//   * multi-tensor global gradient norm + clip coefficient + AdamW (+ optional EMA lerp) over a device pointer table
//     (training/optimization/build.py:86-130, trainer.py:321 gradient_clip_val, callbacks.py:869-907 EMA): the clip
//     coefficient stays on the device, so a step needs no host synchronisation.
// HBM bound: 16 B / parameter (20 B with EMA) and 8-12 B / output voxel; tiny next to the network's traffic.
#include "pytc_common.h"

namespace pytc {

struct LossGeom {
  long xs_n, xs_c, xs_r, ts_n, ts_c, ts_r, ws_n, ws_c, ws_r, ds_n, ds_c, ds_r;
  int N, C;
  long R;
  float pos_weight, w_bce, w_dice, smooth_nr, smooth_dr;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// sums[slot][n*C + c][5] = (sum_{w>0} w*bce, #{w>0}, sum_{w>0} p*t, sum_{w>0} p, sum_{w>0} t) over the slot's rows
__global__ void __launch_bounds__(256)
bce_dice_sums_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ w,
                     float* __restrict__ part, LossGeom g, long rows_per_slot) {
  __shared__ float sm[5][256];
  const int slot = blockIdx.x, nc = blockIdx.y;
  const int n = nc / g.C, c = nc % g.C;
  const long r0 = (long)slot * rows_per_slot;
  const long r1 = r0 + rows_per_slot < g.R ? r0 + rows_per_slot : g.R;
  const float* xp = x + n * g.xs_n + c * g.xs_c;
  const float* tp = t + n * g.ts_n + c * g.ts_c;
  const float* wp = w ? w + n * g.ws_n + c * g.ws_c : nullptr;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (long r = r0 + threadIdx.x; r < r1; r += 256) {
    const float xv = xp[r * g.xs_r], tv = tp[r * g.ts_r];
    const float wv = wp ? wp[r * g.ws_r] : 1.0f;
    // torch's binary_cross_entropy_with_logits: (1 - t) x + (1 + (pw - 1) t) (log1p(exp(-|x|)) + max(-x, 0))
    const float lw = 1.0f + (g.pos_weight - 1.0f) * tv;
    const float bce = (1.0f - tv) * xv + lw * (log1pf(__expf(-fabsf(xv))) + fmaxf(-xv, 0.f));
    const float p = sigmoidf_(xv);
    if (wv > 0.f) {
      s[0] += wv * bce; s[1] += 1.0f;      // mean over the valid (weight > 0) voxels, losses.py:17-44
      // Dice has no weight argument: the reference feeds it masked INPUTS (orchestrator.py:648-655: invalid voxels get the
      // clamp floor -20 as logit and 0 as target), i.e. they drop out of all three sums up to sigmoid(-20) = 2e-9 each
      s[2] += p * tv; s[3] += p; s[4] += tv;
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) sm[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) part[((long)slot * gridDim.y + nc) * 5 + threadIdx.x] = sm[threadIdx.x][0];
}

// one workgroup: sums[nc][5] = sum over slots (slot order); out = (loss, bce term, dice term, bce denominator)
__global__ void __launch_bounds__(256)
bce_dice_finalize_kernel(const float* __restrict__ part, float* __restrict__ sums, float* __restrict__ out, LossGeom g,
                         int slots, int has_w) {
  __shared__ float sb[256], sw[256], sd[256];
  const int NC = g.N * g.C;
  float ab = 0.f, aw = 0.f, ad = 0.f;
  for (int nc = threadIdx.x; nc < NC; nc += 256) {
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < slots; ++sl)
#pragma unroll
      for (int k = 0; k < 5; ++k) s[k] += part[((long)sl * NC + nc) * 5 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) sums[nc * 5 + k] = s[k];
    ab += s[0]; aw += s[1];
    ad += 1.0f - (2.0f * s[2] + g.smooth_nr) / (s[3] + s[4] + g.smooth_dr);
  }
  sb[threadIdx.x] = ab; sw[threadIdx.x] = aw; sd[threadIdx.x] = ad;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      sb[threadIdx.x] += sb[threadIdx.x + st]; sw[threadIdx.x] += sw[threadIdx.x + st]; sd[threadIdx.x] += sd[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float den = has_w ? fmaxf(sw[0], 1.0f) : (float)NC * (float)g.R;      // no valid voxel: bce = 0 / 1
    const float bce = sb[0] / den, dice = sd[0] / (float)NC;
    out[0] = g.w_bce * bce + g.w_dice * dice; out[1] = bce; out[2] = dice; out[3] = den;
  }
}

// dx = gout * ( w_bce * w/den * (p (1 + (pw-1) t) - pw t) + w_dice/(N C) * ((2 I + e_nr) - 2 t Dn) / Dn^2 * p (1-p) )
__global__ void __launch_bounds__(256)
bce_dice_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ w,
                    const float* __restrict__ sums, const float* __restrict__ out, const float* __restrict__ gout,
                    float* __restrict__ dx, LossGeom g) {
  const int nc = blockIdx.y;
  const int n = nc / g.C, c = nc % g.C;
  const float go = gout[0];
  const float kb = go * g.w_bce / out[3];
  const float I2 = 2.0f * sums[nc * 5 + 2] + g.smooth_nr, Dn = sums[nc * 5 + 3] + sums[nc * 5 + 4] + g.smooth_dr;
  const float kd = go * g.w_dice / ((float)(g.N * g.C) * Dn * Dn);
  const float* xp = x + n * g.xs_n + c * g.xs_c;
  const float* tp = t + n * g.ts_n + c * g.ts_c;
  const float* wp = w ? w + n * g.ws_n + c * g.ws_c : nullptr;
  float* dp = dx + n * g.ds_n + c * g.ds_c;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < g.R; r += (long)gridDim.x * 256) {
    const float xv = xp[r * g.xs_r], tv = tp[r * g.ts_r];
    const float wv = wp ? wp[r * g.ws_r] : 1.0f;
    const float p = sigmoidf_(xv);
    const float gb = wv > 0.f ? kb * wv * (p * (1.0f + (g.pos_weight - 1.0f) * tv) - g.pos_weight * tv) : 0.f;
    const float gd = wv > 0.f ? kd * (I2 - 2.0f * tv * Dn) * p * (1.0f - p) : 0.f;      // masked_fill blocks the gradient
    dp[r * g.ds_r] = gb + gd;
  }
}

// ---- multi-tensor optimizer -------------------------------------------------------------------------------------------
struct OptTensor { float* p; const float* g; float* m; float* v; float* ema; long n; int group; int pad; };
struct OptGroup { float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, ema_decay; };
constexpr int OPT_CHUNK = 4096;     // elements per workgroup
constexpr int OPT_MAX_GROUPS = 8;
struct OptGroups { OptGroup g[OPT_MAX_GROUPS]; };   // passed by value: the per-step scalars need no H2D copy

// part[chunk] = sum g^2 over the chunk (chunk -> (tensor, offset) through the table)
__global__ void __launch_bounds__(256)
grad_sqnorm_kernel(const OptTensor* __restrict__ tab, const int2* __restrict__ chunks, float* __restrict__ part) {
  __shared__ float sm[256];
  const int2 ck = chunks[blockIdx.x];
  const OptTensor T = tab[ck.x];
  const long o0 = (long)ck.y * OPT_CHUNK;
  const long o1 = o0 + OPT_CHUNK < T.n ? o0 + OPT_CHUNK : T.n;
  float s = 0.f;
  for (long i = o0 + threadIdx.x; i < o1; i += 256) { const float gv = T.g[i]; s = fmaf(gv, gv, s); }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

// norm_coef[0] = ||g||_2, norm_coef[1] = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ void __launch_bounds__(256)
grad_norm_finalize_kernel(const float* __restrict__ part, int nchunks, float max_norm, float* __restrict__ norm_coef) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nchunks; i += 256) s += part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(sm[0]);
    norm_coef[0] = nrm;
    const float cf = max_norm > 0.f ? max_norm / (nrm + 1e-6f) : 1.0f;
    norm_coef[1] = cf < 1.0f ? cf : 1.0f;
  }
}

// torch.optim.AdamW (decoupled decay): p *= 1 - lr wd; m, v moments of coef*g; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void __launch_bounds__(256)
adamw_multi_kernel(const OptTensor* __restrict__ tab, const int2* __restrict__ chunks, OptGroups groups,
                   const float* __restrict__ norm_coef) {
  const int2 ck = chunks[blockIdx.x];
  const OptTensor T = tab[ck.x];
  const OptGroup G = groups.g[T.group];
  const float coef = norm_coef ? norm_coef[1] : 1.0f;
  const long o0 = (long)ck.y * OPT_CHUNK;
  const long o1 = o0 + OPT_CHUNK < T.n ? o0 + OPT_CHUNK : T.n;
  const float step = G.lr / G.bc1;
  for (long i = o0 + threadIdx.x; i < o1; i += 256) {
    const float gv = T.g[i] * coef;
    float pv = T.p[i] * (1.0f - G.lr * G.wd);
    const float mv = G.beta1 * T.m[i] + (1.0f - G.beta1) * gv;
    const float vv = G.beta2 * T.v[i] + (1.0f - G.beta2) * gv * gv;
    pv -= step * mv / (sqrtf(vv) / G.bc2_sqrt + G.eps);
    T.p[i] = pv; T.m[i] = mv; T.v[i] = vv;
    if (T.ema) T.ema[i] = G.ema_decay * T.ema[i] + (1.0f - G.ema_decay) * pv;
  }
}

}  // namespace pytc

using namespace pytc;

static int loss_slots(long R) {
  const long s = R / 16384;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

extern "C" int64_t pytc_bce_dice_ws_elems(int N, int C, int64_t R) {
  if (N < 1 || C < 1 || R < 1) return -1;
  return (int64_t)loss_slots(R) * N * C * 5;
}

static LossGeom make_geom(int N, int C, int64_t R, const int64_t* xs, const int64_t* ts, const int64_t* ws,
                          const int64_t* ds, float pos_weight, float w_bce, float w_dice, float snr, float sdr) {
  LossGeom g{};
  g.xs_n = xs[0]; g.xs_c = xs[1]; g.xs_r = xs[2];
  g.ts_n = ts[0]; g.ts_c = ts[1]; g.ts_r = ts[2];
  if (ws) { g.ws_n = ws[0]; g.ws_c = ws[1]; g.ws_r = ws[2]; }
  if (ds) { g.ds_n = ds[0]; g.ds_c = ds[1]; g.ds_r = ds[2]; }
  g.N = N; g.C = C; g.R = R;
  g.pos_weight = pos_weight; g.w_bce = w_bce; g.w_dice = w_dice; g.smooth_nr = snr; g.smooth_dr = sdr;
  return g;
}

extern "C" int pytc_bce_dice_fwd(const float* logits, const float* target, const float* weight, int N, int C, int64_t R,
                                 const int64_t* x_strides, const int64_t* t_strides, const int64_t* w_strides,
                                 float pos_weight, float w_bce, float w_dice, float smooth_nr, float smooth_dr,
                                 float* workspace, float* sums, float* out, void* stream) {
  PYTC_REQUIRE(logits && target && workspace && sums && out && x_strides && t_strides, "bce_dice_fwd: null pointer");
  PYTC_REQUIRE(N >= 1 && C >= 1 && R >= 1 && (long)N * C <= 65535, "bce_dice_fwd: bad shape");
  PYTC_REQUIRE(!weight || w_strides, "bce_dice_fwd: weight without strides");
  const LossGeom g = make_geom(N, C, R, x_strides, t_strides, weight ? w_strides : nullptr, nullptr, pos_weight, w_bce,
                               w_dice, smooth_nr, smooth_dr);
  const int slots = loss_slots(R);
  const long rps = (R + slots - 1) / slots;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bce_dice_sums_kernel, dim3(slots, N * C), dim3(256), 0, s, logits, target, weight, workspace, g, rps);
  hipLaunchKernelGGL(bce_dice_finalize_kernel, dim3(1), dim3(256), 0, s, workspace, sums, out, g, slots, weight ? 1 : 0);
  PYTC_LAUNCH_CHECK("bce_dice_fwd");
  return PYTC_OK;
}

extern "C" int pytc_bce_dice_bwd(const float* logits, const float* target, const float* weight, const float* sums,
                                 const float* out, const float* grad_out, float* dlogits, int N, int C, int64_t R,
                                 const int64_t* x_strides, const int64_t* t_strides, const int64_t* w_strides,
                                 const int64_t* d_strides, float pos_weight, float w_bce, float w_dice, float smooth_nr,
                                 float smooth_dr, void* stream) {
  PYTC_REQUIRE(logits && target && sums && out && grad_out && dlogits && x_strides && t_strides && d_strides,
               "bce_dice_bwd: null pointer");
  PYTC_REQUIRE(N >= 1 && C >= 1 && R >= 1 && (long)N * C <= 65535, "bce_dice_bwd: bad shape");
  const LossGeom g = make_geom(N, C, R, x_strides, t_strides, weight ? w_strides : nullptr, d_strides, pos_weight, w_bce,
                               w_dice, smooth_nr, smooth_dr);
  long bx = (R + 1023) / 1024;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(bce_dice_bwd_kernel, dim3((unsigned)bx, N * C), dim3(256), 0, (hipStream_t)stream, logits, target,
                     weight, sums, out, grad_out, dlogits, g);
  PYTC_LAUNCH_CHECK("bce_dice_bwd");
  return PYTC_OK;
}

extern "C" int pytc_opt_chunk_elems(void) { return OPT_CHUNK; }

// table: n_tensors records of 6 int64 (p, g, m, v, ema, n) + group index as the 7th; chunks: n_chunks (tensor, chunk) pairs
extern "C" int pytc_grad_norm_multi(const void* table, const void* chunks, int n_chunks, float max_norm, float* workspace,
                                    float* norm_coef, void* stream) {
  PYTC_REQUIRE(table && chunks && workspace && norm_coef && n_chunks >= 1, "grad_norm_multi: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, s, (const OptTensor*)table, (const int2*)chunks, workspace);
  hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(256), 0, s, workspace, n_chunks, max_norm, norm_coef);
  PYTC_LAUNCH_CHECK("grad_norm_multi");
  return PYTC_OK;
}

extern "C" int pytc_adamw_multi(const void* table, const void* chunks, int n_chunks, const float* groups_host, int n_groups,
                                const float* norm_coef, void* stream) {
  PYTC_REQUIRE(table && chunks && groups_host && n_chunks >= 1, "adamw_multi: bad arguments");
  PYTC_REQUIRE(n_groups >= 1 && n_groups <= OPT_MAX_GROUPS, "adamw_multi: 1..8 parameter groups");
  OptGroups G{};
  for (int i = 0; i < n_groups; ++i) {
    const float* q = groups_host + 8 * i;
    G.g[i] = OptGroup{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]};
  }
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)table,
                     (const int2*)chunks, G, norm_coef);
  PYTC_LAUNCH_CHECK("adamw_multi");
  return PYTC_OK;
}
