// Shared pieces of the pointwise (1x1x1) MFMA kernels: operand traits and the residual/store epilogue.
#pragma once
#include "pytc_common.h"

namespace pytc {

template <typename TW>
struct Mma;

template <>
struct Mma<bf16_t> {
  static constexpr int EPL = 8;     // fragment elements per lane
  static constexpr int KSTEP = 32;  // k covered by one fragment set
  typedef bf16x8_t frag_t;
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag_t from_floats(const float (&v)[8]) {
    f32x8_t f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = v[i];
    return __builtin_convertvector(f, frag_t);
  }
};

template <>
struct Mma<float> {
  static constexpr int EPL = 4;
  static constexpr int KSTEP = 16;
  typedef f32x4_t frag_t;
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    // logical k of (step s, lane group kb) = kb*4 + s : any bijection works as long as A and B agree
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ frag_t from_floats(const float (&v)[4]) {
    frag_t f;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
    return f;
  }
};

// Row permutation used by the "paired" packing: two consecutive 16-row tiles (T = 2p, 2p+1) are
// interleaved so that the MFMA C/D layout (lane group kb owns rows 4kb..4kb+3 of each tile) leaves
// every lane with 8 CONSECUTIVE logical rows p*32 + kb*8 + 0..7:
//     logical(T, m) = (T>>1)*32 + (m>>2)*8 + (T&1)*4 + (m&3)
// -> 16-byte NDHWC stores / residual loads, 8 consecutive biases, and a GEMM1 accumulator that IS the
//    natural-k-order B operand of GEMM2 (no cross-lane movement between the two GEMMs).
__host__ __device__ __forceinline__ int paired_row(int T, int m) {
  return (T >> 1) * 32 + (m >> 2) * 8 + (T & 1) * 4 + (m & 3);
}

struct EpiParams {
  const void* res;        // RES_ADD: residual [rows][C_out]; RES_UPSAMPLE: encoder skip [rows][C_out]
  const void* res_low;    // RES_UPSAMPLE: low-res transposed-1x1 residual (bias included)
  const float* res_bias;  // RES_UPSAMPLE: bias of that residual conv (value in the stride holes)
  void* y;
  long rps_out;           // output rows per sample
  int C_out;
  int res_mode;
  int Go_d, Go_h, Go_w;   // RES_UPSAMPLE: output grid
  int Gl_d, Gl_h, Gl_w;   // RES_UPSAMPLE: low-res grid
  // 1: the big streams (operand rows, residual / skip rows, output rows) as nontemporal loads / stores (knob stream_nt; measured
  // neutral for the mixers, see stream_nt_policy)
  int nt;
};

// 16-byte row piece, plain or nontemporal (wave-uniform flag)
typedef unsigned int pytc_u32x4_t __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ V ld_stream(const V* ptr, int nt) {
  static_assert(sizeof(V) == 16, "16-byte row pieces");
  if (nt) return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const pytc_u32x4_t*>(ptr)));
  return *ptr;
}

// policy of the launch wrappers: knob stream_nt = 1 puts the mixers' big streams on nontemporal loads / stores, 0 (default) keeps plain
// ones.  Measured (profiles/r05_stream_policy.txt): a bare three-stream kernel gains 9 % from nt (5.9 -> 6.45 TB/s), the mixers gain
// nothing (32->64->32: 543 vs 536 us isolated, whole step 6.47 vs 6.48 ms per 8 windows) -- they saturate the memory system at ~5 TB/s
// with ~4 us of queueing latency whatever the cache policy -- so the default stays plain.
static inline int stream_nt_policy(long bytes_largest_tensor) {
  (void)bytes_largest_tensor;
  return tuning_get("stream_nt", 0) > 0 ? 1 : 0;
}

// v[NCH] = conv result (+bias, activation) for channels o0..o0+NCH-1 of output row `orow` of sample n.
// GELU_BWD = false compiles the PYTC_RES_GELU_BWD branch out (the fused mixer never uses it and pays registers for it).
// pos != nullptr (RES_UPSAMPLE only): {pz, py, px} of `orow` in the output grid, already known to the caller (the fused mixer
// derives them once per wave and steps them per tile: the generic form below costs two integer divisions per stored 16 bytes).
// COUT > 0: the caller knows C_out at compile time (the fused mixers: MO * 16) -- row offsets become shifts instead of 64-bit multiplies
// (v_mad_u64_u32 / v_mul_lo_u32 are quarter rate, and instruction time adds to memory time on this machine: DESIGN.md 4.17).
template <typename TO, int NCH, bool GELU_BWD = false, int COUT = 0>
__device__ __forceinline__ void finish_and_store(float (&v)[NCH], const EpiParams& e, int n, long orow, int o0,
                                                 const float* pre = nullptr, const int* pos = nullptr) {
  const int C_out = COUT > 0 ? COUT : e.C_out;
  // pre != nullptr: the caller already loaded res[orow][o0..o0+NCH) (prefetched ahead of the GEMMs)
  TO* yn = reinterpret_cast<TO*>(e.y) + (long)n * e.rps_out * C_out;
  const TO* resn = e.res ? reinterpret_cast<const TO*>(e.res) + (long)n * e.rps_out * C_out : nullptr;
  const long off = orow * C_out + o0;
  const bool full = (C_out % NCH) == 0 && (o0 + NCH <= C_out);
  if (e.res_mode == PYTC_RES_ADD) {
    float rv[NCH];
    if (pre) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rv[i] = pre[i];
    } else if (full) VecIO<TO, NCH>::load(resn + off, rv);
    else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rv[i] = (o0 + i < C_out) ? to_f32<TO>(resn[off + i]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] += rv[i];
  } else if (GELU_BWD && e.res_mode == PYTC_RES_GELU_BWD) {
    float rv[NCH];
    if (pre) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rv[i] = pre[i];
    } else if (full) VecIO<TO, NCH>::load(resn + off, rv);
    else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rv[i] = (o0 + i < C_out) ? to_f32<TO>(resn[off + i]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] *= gelu_erf_grad(rv[i]);
  } else if (GELU_BWD && e.res_mode == PYTC_RES_NORM_BWD) {
    // GroupNorm backward of the GEMM's own (unrounded) result: dt = A*v + B*t + C per (sample, channel), R = t
    float rv[NCH];
    if (pre) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rv[i] = pre[i];
    } else if (full) VecIO<TO, NCH>::load(resn + off, rv);
    else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rv[i] = (o0 + i < C_out) ? to_f32<TO>(resn[off + i]) : 0.f;
    }
    const float* cf = e.res_bias + (long)n * 3 * C_out + o0;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (o0 + i < C_out) v[i] = fmaf(cf[i], v[i], fmaf(cf[C_out + i], rv[i], cf[2 * C_out + i]));
    if (e.Go_w > 0) {
      // up blocks (round 6): the rows are the padded (Go_d, Go_h, Go_w) grid of a transposed conv's output, the front faces are not part
      // of it -- dt goes to the compact (Go_d - 1, Go_h - 1, Go_w - 1) grid the transposed conv's backward reads, face rows are dropped
      const unsigned ur = (unsigned)orow, gw = (unsigned)e.Go_w, gh = (unsigned)e.Go_h;
      const unsigned tq = ur / gw;
      const int px = (int)(ur - tq * gw), pz = (int)(tq / gh), py = (int)(tq - (unsigned)pz * gh);
      if (px == 0 || py == 0 || pz == 0) return;
      const long crow = ((long)(pz - 1) * (e.Go_h - 1) + (py - 1)) * (e.Go_w - 1) + (px - 1);
      TO* yc = reinterpret_cast<TO*>(e.y) + ((long)n * (e.Go_d - 1) * (e.Go_h - 1) * (e.Go_w - 1) + crow) * C_out + o0;
      if (full) VecIO<TO, NCH>::store(yc, v);
      else {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          if (o0 + i < C_out) yc[i] = from_f32<TO>(v[i]);
      }
      return;
    }
  } else if (e.res_mode == PYTC_RES_UPSAMPLE) {
    int px, py, pz;
    if (pos) { pz = pos[0]; py = pos[1]; px = pos[2]; }
    else {
      // rows per sample fit 32 bits (checked at the API): 32-bit divisions -- the 64-bit forms compile to a call-sized sequence
      const unsigned ur = (unsigned)orow, gw = (unsigned)e.Go_w, gh = (unsigned)e.Go_h;
      const unsigned t = ur / gw;
      px = (int)(ur - t * gw);
      pz = (int)(t / gh);
      py = (int)(t - (unsigned)pz * gh);
    }
    float sk[NCH];
    if (pre) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) sk[i] = pre[i];
    } else if (full) VecIO<TO, NCH>::load(resn + off, sk);
    else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) sk[i] = (o0 + i < C_out) ? to_f32<TO>(resn[off + i]) : 0.f;
    }
    if (px == 0 || py == 0 || pz == 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) v[i] = sk[i];
    } else {
      const int oz = pz - 1, oy = py - 1, ox = px - 1;
      float rl[NCH];
      if (e.res_low && !((oz | oy | ox) & 1)) {
        const TO* rp = reinterpret_cast<const TO*>(e.res_low) +
                       ((((long)n * e.Gl_d + (oz >> 1)) * e.Gl_h + (oy >> 1)) * e.Gl_w + (ox >> 1)) * C_out + o0;
        if (full) VecIO<TO, NCH>::load(rp, rl);
        else {
#pragma unroll
          for (int i = 0; i < NCH; ++i) rl[i] = (o0 + i < C_out) ? to_f32<TO>(rp[i]) : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NCH; ++i) rl[i] = (e.res_bias && o0 + i < C_out) ? e.res_bias[o0 + i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NCH; ++i) v[i] = v[i] + rl[i] + sk[i];
    }
  }
  if (full) {
    if constexpr (sizeof(TO) == 2 && NCH == 8) {
      if (e.nt) {          // same conversion as VecIO::store, nontemporal 16-byte store
        f32x8_t f;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = v[i];
        __builtin_nontemporal_store(__builtin_convertvector(f, bf16x8_t), reinterpret_cast<bf16x8_t*>(yn + off));
        return;
      }
    }
    VecIO<TO, NCH>::store(yn + off, v);
  } else {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (o0 + i < C_out) yn[off + i] = from_f32<TO>(v[i]);
  }
}

template <typename TI, int EPL>
__device__ __forceinline__ void load_row_frag(const TI* __restrict__ row, int k0, int C_in, bool vec_ok,
                                              float (&v)[EPL]) {
  if (vec_ok && k0 + EPL <= C_in) {
    if constexpr (sizeof(TI) == 4 && EPL == 8) {
      float t0[4], t1[4];
      VecIO<float, 4>::load(reinterpret_cast<const float*>(row) + k0, t0);
      VecIO<float, 4>::load(reinterpret_cast<const float*>(row) + k0 + 4, t1);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = t0[i]; v[4 + i] = t1[i]; }
    } else {
      VecIO<TI, EPL>::load(row + k0, v);
    }
  } else {
#pragma unroll
    for (int j = 0; j < EPL; ++j) v[j] = (k0 + j < C_in) ? to_f32<TI>(row[k0 + j]) : 0.f;
  }
}

// L2 warm-up for weight streams: inside a network every launch meets its weights cold, and a wave's hidden-chunk loop
// is a chain of dependent weight loads (one HBM round trip per link).  At kernel entry the workgroups of each XCD
// (dispatch order: linear workgroup id % 8) together touch every 128-byte line of the image once, so the chain runs
// against L2 hits instead.  Fire-and-forget loads into a scratch register; no synchronisation is involved.
// The destination registers stay reserved until warm_l2_done() -- a load into a register the compiler believes dead
// would overwrite whatever it keeps there when the data arrives.
template <int PER>                      // PER loads per image per lane; two images (w2, w3) at most
struct WarmRegs { unsigned int r[2 * PER]; };
template <int PER>
__device__ __forceinline__ void warm_l2(const void* base, long bytes, WarmRegs<PER>& w, int first) {
  const long lines = (bytes + 127) >> 7;
  const long lin = (long)blockIdx.z * gridDim.y * gridDim.x + (long)blockIdx.y * gridDim.x + blockIdx.x;
  const long total = (long)gridDim.x * gridDim.y * gridDim.z;
  const long per_xcd = (total + 7) >> 3;
  const long stride = per_xcd * blockDim.x;
  long l = (lin >> 3) * blockDim.x + threadIdx.x;
#pragma unroll
  for (int it = 0; it < PER; ++it, l += stride) {
    const long lc = l < lines ? l : lines - 1;          // always issue (clamped): the register is written either way
    const char* ptr = reinterpret_cast<const char*>(base) + (lc << 7);
    asm volatile("global_load_dword %0, %1, off" : "=v"(w.r[first + it]) : "v"(ptr) : "memory");
  }
}
template <int PER>
__device__ __forceinline__ void warm_l2_done(WarmRegs<PER>& w) {
  // explicit drain, then release the registers: the compiler's own wait for the operand loads may be scheduled later
  // than this point, so it cannot be relied on.  Call this where the operand tile is consumed anyway.
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
  for (int i = 0; i < 2 * PER; ++i) asm volatile("" : : "v"(w.r[i]) : "memory");
}

__device__ __forceinline__ float apply_act(float t, int act) {
  if (act == PYTC_ACT_GELU) return gelu_erf(t);
  if (act == PYTC_ACT_SIGMOID) return 1.f / (1.f + __expf(-t));
  if (act == PYTC_ACT_TANH) return tanhf(t);
  return t;
}

}  // namespace pytc
