// Depthwise 3x3x3 convolution (stride 1, bf16 NDHWC) on the MATRIX cores: v_mfma_f32_4x4x4_16b_bf16 computes SIXTEEN independent
// 4x4x4 products per instruction, and a depthwise conv shares nothing between channels -- so one block of the instruction is one
// CHANNEL (round 4; the 16x16x32 Toeplitz form of round 3/4, one channel per instruction with 29/32 zeros, lost 2.3x to the VALU
// z-march: profiles/r04_toeplitz_probe.txt).  Per block (= channel c) and per in-plane row offset dy:
//     D[i][j] += sum_k A[i][k] * B[k][j]      i = output plane gz - 1 + i (z tap dz = 2 - i), k = input COLUMN, j = output row
//     A[i][k] = w_c[dz = 2 - i][dy][dx = k - r]   (r = 0 / 1: the two output columns that share the four input columns; row i = 3 is zero)
//     B[k][j] = in_c[gz][y0 + j + dy][x0 + k]     (four consecutive columns of one row of the haloed plane, read as 8 bytes from an LDS
//                                                  image that keeps x innermost)
// so one instruction retires 16 channels x 3 z taps x 3 x taps x 4 rows = 576 useful MACs (56 % of its 1024), and the whole stencil
// of (16 channels x 4 rows x 2 columns x 1 input plane) is 6 instructions (12 with the hi/lo weight split) against 108 v_pk_fma_f16 +
// 63 LDS reads for the same work in the z-march.  The z extent rides in the accumulator's four VGPRs (outputs gz-1, gz, gz+1, unused):
// after the instructions of input plane gz VGPR 0 is a finished output and the tuple rotates.
// Round 5 (profiles/r05_additivity.txt): the round-4 kernel kept y innermost (k = input row).  rocprofv3 showed it LDS bound -- LDS
// pipe 88 % busy, 60 % of those cycles bank conflicts -- and tools/lds_conflict_model.py reproduces both figures from the address
// algebra: 848 LDS-array cycles per workgroup step (4 workgroups per CU: 3 392 against a measured step of 3 339 cycles), of which the
// 2-byte commit writes were 416 (the four 8-channel parts of a voxel, 8 x 120 halfwords apart, all fell on ONE bank: 4-way), the
// operand reads 288 and the tile writes 128 -- "memory time and instruction time add"  was this queue.  With x
// innermost the lanes of a commit write walk the innermost axis (as they walk global memory), the channel stride is an odd number
// of dwords (parts -> banks 0 / 8 / 16 / 24), rows are 8 dwords apart (operand reads: 32 lanes -> 32 banks) and the tile rows are
// padded by 4 dwords: 376 cycles per workgroup step, no conflict left.
// Operands: activations are bf16 as stored (no conversion, no range clamp); the fp32 weights go in as bf16 -- hi halves only by default
// (= the weights torch.autocast hands the reference's Conv3d), hi + lo pairs on request (two instructions per operand, 16 mantissa
// bits: fp32-weight accuracy) -- and accumulation is fp32 throughout (the packed-f16 z-march sums nine taps in f16).
// Data movement is the z-march's: 8 x 8 footprint of a 32-channel group, z-chunks, one haloed plane per step through LDS (double
// buffered), plane loads as inline asm with counted waits (PF planes in flight in registers), HBM sees x once (+ halo) and y once,
// statistics leave as one partial per workgroup.  Both transpositions (NDHWC <-> channel-major rows) happen in LDS: 2-byte writes at
// commit, and a 4 KB tile per workgroup that turns the accumulator layout (lane = channel x column) back into 16-byte NDHWC stores
// (full 64-byte voxel rows, one step later).
// Measured and removed (profiles/r04_dwconv_mfma.txt; level 0, 8 windows, hi + lo: 435 us): (a) a FIFTH wave that only stores, so that
// the other waves' vmcnt (loads and stores share it on gfx9) counts loads only: 555 us (3 workgroups per CU instead of 4); (b) waves
// SPECIALISED into 4 compute + 4 loader waves (512 threads, 6-8 planes in flight): 436 us, and its 3-planes build faulted (an asm-issued
// load landing in a register the allocator had meanwhile reused -- the counted-wait scheme tolerates no live-range split, see build.py);
// (c) 2 / 3 planes in flight: no difference.  Timing probes of this kernel: loads + commit only 160-180 us, + output path 310 us (the
// copy-speed roof of this footprint: 1.2x input halo), + hi instructions 360-370 us, + lo instructions 435 us -- the matrix work is
// NOT hidden behind the memory time although the pipe is 40 % busy; the VALU z-march it replaces runs at 430-490 us.
#include <type_traits>

#include "dwconv_march.h"
#include "pw_common.h"

namespace pytc {

typedef short s4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2a4_t __attribute__((ext_vector_type(2), aligned(4)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

constexpr int MF_TY = 8, MF_TX = 8, MF_CG = 32;
constexpr int MF_EY = MF_TY + 2, MF_EX = MF_TX + 2;
constexpr int MF_EXP = 16;                              // halfwords per (channel, row) line of the image: 10 columns + pad (8 dwords)
constexpr int MF_CS = MF_EY * MF_EXP + 2;               // halfwords per channel: 81 dwords (odd: 8 channels on = 8 banks on)
constexpr int MF_IMG = MF_CG * MF_CS;                   // halfwords per staged plane (10.1 KB)
constexpr int MF_NCHUNK = MF_EY * MF_EX * (MF_CG / 8);  // 16-byte chunks per plane
constexpr int MF_CPT = 2;                               // chunks per thread (256 threads)
constexpr int MF_TS = 32;                               // output tile: halfwords per position (32 channels, NDHWC)
constexpr int MF_RS = MF_TX * MF_TS + 8;                // ... per footprint row: + 4 dwords (the four rows of a unit -> banks 0 / 4 / 8 / 12)

__device__ __forceinline__ unsigned short mf_bf16_bits(float f) {       // round to nearest even
  unsigned int u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float mf_bf16_float(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

template <int N>
__device__ __forceinline__ void mf_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// PF: planes in flight (register staged).  LO: second instruction per operand with the low halves of the weights.
// PROBE (timing probes, wrong results): 1 = no matrix instructions, 3 = no matrix instructions and no output path;
// 4 = right results, and instead of the statistics every wave leaves the shader cycles (s_memtime) it spent in each section of its
// plane steps -- stats[(n, slot, 0, cg*32 + wave*8 + k)], k = 0 flush + load issue, 1 matrix instructions, 2 wait for the staged
// plane, 3 LDS commit, 4 output rounding + tile writes, 5 barrier, 6 steps, 7 whole kernel (profiles/r05_additivity.txt).
// STORE = false: the statistics-only pass of the fused block (pw_dwmix_kernels.hip): same products, same rounding, same partial sums
// in the same order as the storing kernel (bit-identical statistics) -- the output tile, its LDS traffic and the HBM stores are gone.
// MIXHC > 0: the FUSED BLOCK (round 5; C = 32, C_out = 32, hidden width 32 * MIXHC): the depthwise output of a plane goes no
// further than the workgroup's LDS tile -- a step later every wave reads 16 of its 64 positions back as the B operand of the
// expanding 1x1x1 conv and runs the channel mixer of pw_mlp_kernels.hip on them (GroupNorm folded into per-sample expand weights by
// groupnorm_fold_mlp_kernel from the statistics of a STORE = false pass; packed-fp16 GELU; f16 projection; + x; one 16-byte store
// per lane).  Same products, same accumulation order, same roundings as dwconv3d_k3_mfma_kernel followed by pw_mlp_kernel<1, 2, NT, 3>
// with per-sample operands: the block output is BIT-IDENTICAL to the two-launch path, and the 2 * C bytes per voxel of the depthwise
// tensor never reach HBM (read x twice, write y once).  The residual rows ride in the asm-load pipeline: R(z) = the
// 64 x 64 bytes of x at the footprint of plane z, requested at step z (before the plane loads of that step), consumed at step z + 2.

template <int N>
__device__ __forceinline__ void mf_wait_vm_upto(int n) {     // s_waitcnt vmcnt(n) for a wave-uniform runtime n in 0..N
  if constexpr (N > 0) {
    if (n >= N) { mf_wait_vm<N>(); return; }
    mf_wait_vm_upto<N - 1>(n);
  } else {
    mf_wait_vm<0>();
  }
}

template <int PF, bool LO, int PROBE = 0, bool STORE = true, int MIXHC = 0>
__global__ void __launch_bounds__(256, MIXHC ? 3 : 4)
dwconv3d_k3_mfma_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, const float* __restrict__ w,
                        const float* __restrict__ bias, float* __restrict__ stats, DwMarch g, DwMix mx) {
  static_assert(PF == 2 || PF == 3, "two or three planes in flight");
  static_assert(MIXHC == 0 || (STORE && PF == 3), "the fused block keeps the output tile and three planes in flight");
  constexpr bool MIX = MIXHC > 0;
  __shared__ __attribute__((aligned(16))) bf16x8_t w2s[MIX ? MIXHC * 2 * 64 : 1];
  __shared__ __attribute__((aligned(16))) h8_t w3s[MIX ? 2 * MIXHC * 64 : 1];
  __shared__ __attribute__((aligned(16))) float mbias[MIX ? 32 * MIXHC + 32 : 4];     // folded expand bias of this sample | projection bias
  __shared__ __attribute__((aligned(16))) unsigned short image[2][MF_IMG];
  __shared__ __attribute__((aligned(16))) unsigned short otile[STORE ? 2 : 1][STORE ? MF_TY * MF_RS : 8];   // per step parity: 8 rows x (8 positions x 32 channels + pad)
  __shared__ float wl[27 * MF_CG];
  __shared__ float red[4][2][16];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int b = g.swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  // channel groups innermost: the workgroups that split a voxel's channels (64 bytes each of its 128-byte lines) are dispatched back
  // to back on one XCD, so the second one finds the lines in L2 (slot-major order fetched every line of a C >= 64 level twice:
  // 56^3 x 64 at 2.4 TB/s against 3.4 at C = 32, profiles/r04_dwconv_mfma.txt)
  const int ncg = g.C / MF_CG;
  const int cg = g.cg_inner ? b % ncg : (b / g.slots) % ncg;
  const int slot_id = g.cg_inner ? (b / ncg) % g.slots : b % g.slots;
  const int n = b / (ncg * g.slots);
  b = slot_id;
  const int fx = b % g.tx; b /= g.tx;
  const int fy = b % g.ty;
  const int zchunk = b / g.ty;
  const int y0 = fy * MF_TY, x0 = fx * MF_TX;
  const int zs = zchunk * g.zc;
  const int ze = min(zs + g.zc, g.D);               // outputs [zs, ze)
  const int C = g.C;
  const long plane_elems = (long)g.H * g.W * C;
  const unsigned short* xn = x + (long)n * g.D * plane_elems + cg * MF_CG;
  unsigned short* yn = y + (long)n * g.D * plane_elems + cg * MF_CG;
  // footprint with its halo inside the volume in y and x: no lane of this workgroup ever masks anything in-plane (wave-uniform)
  const bool inner = y0 >= 1 && y0 + MF_TY + 1 <= g.H && x0 >= 1 && x0 + MF_TX + 1 <= g.W;

  // ---- weights: 27 x 32 taps through LDS
  for (int i = tid; i < 27 * MF_CG; i += 256) wl[i] = w[(long)(i / MF_CG) * C + cg * MF_CG + (i % MF_CG)];
  if constexpr (MIX) {     // both mixer images of this sample: 4 KB + 4 KB at MIXHC = 2
    const bf16x8_t* w2g = mx.w2n + (long)n * mx.w2_stride;
    for (int i = tid; i < MIXHC * 2 * 64; i += 256) { w2s[i] = w2g[i]; w3s[i] = mx.w3[i]; }
    if (tid < 32 * MIXHC) mbias[tid] = mx.b2n[(long)n * (32 * MIXHC) + tid];
    else if (tid < 32 * MIXHC + 32) mbias[tid] = mx.b3[tid - 32 * MIXHC];
  }
  __syncthreads();

  // ---- staging descriptors (constant along z): chunk = 8 channels of one haloed voxel, consecutive lanes walk channels, then x, then y
  int goff[MF_CPT], loff[MF_CPT];
  bool cok[MF_CPT];
#pragma unroll
  for (int i = 0; i < MF_CPT; ++i) {
    const int c = tid + 256 * i;
    const int vox = c >> 2, part = c & 3;
    const int yy = vox / MF_EX, xx = vox % MF_EX;
    const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
    cok[i] = (c < MF_NCHUNK) && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
    const int gyc = min(max(gy, 0), g.H - 1), gxc = min(max(gx, 0), g.W - 1);
    goff[i] = (gyc * g.W + gxc) * C + part * 8;
    loff[i] = (c < MF_NCHUNK) ? (part * 8) * MF_CS + yy * MF_EXP + xx : -1;       // halfword index of channel part*8 (+q: + q*CS)
  }
  u32x4_t stg[PF][MF_CPT];
  // every lane always loads (clamped address, zero-filled at commit): exactly MF_CPT loads per wave and plane keep the counted waits exact
  auto issue = [&](int gz, u32x4_t (&st)[MF_CPT]) {
    const int zc = min(max(gz, 0), g.D - 1);
#pragma unroll
    for (int i = 0; i < MF_CPT; ++i) {
      const unsigned short* ptr = xn + (long)zc * plane_elems + goff[i];
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(st[i]) : "v"(ptr) : "memory");
    }
  };
  // loads return in order among loads: once at most MF_CPT * (planes requested after the awaited one) operations are outstanding it
  // has landed, whatever the stores (same counter) do
  // `younger`: LOADS requested after the awaited plane (plane loads of later planes, and the fused block's residual loads)
  auto landed = [&](u32x4_t (&st)[MF_CPT], int younger) {
    mf_wait_vm_upto<2 * MF_CPT + 2>(younger);
    asm volatile("" : "+v"(st[0]), "+v"(st[1]) : : "memory");
  };
  auto commit = [&](int slot, u32x4_t (&st)[MF_CPT], int gz) {
    const bool zok = gz >= 0 && gz < g.D;
#pragma unroll
    for (int i = 0; i < MF_CPT; ++i) {
      if (loff[i] < 0) continue;
      unsigned short* dst = &image[slot][loff[i]];
      if (inner && zok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          dst[(2 * q) * MF_CS] = (unsigned short)(st[i][q] & 0xffffu);
          dst[(2 * q + 1) * MF_CS] = (unsigned short)(st[i][q] >> 16);
        }
      } else {
        const bool ok = zok && cok[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned int dw = ok ? st[i][q] : 0u;
          dst[(2 * q) * MF_CS] = (unsigned short)(dw & 0xffffu);
          dst[(2 * q + 1) * MF_CS] = (unsigned short)(dw >> 16);
        }
      }
    }
  };

  // ---- this lane's A operands (constant for the whole march)
  const int h = wave & 1, ph = wave >> 1;             // channel half, row half of the footprint
  const int cl = lane >> 2, j = lane & 3;             // channel within the half (= MFMA block), row within the unit
  const int ch = h * 16 + cl;
  s4_t a_hi[3][2], a_lo[3][2];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      s4_t vh, vl;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = j;                              // A: lane 4b + i holds row i
        const int dz = 2 - i, dx = k - r;
        float wv = 0.f;
        if (i < 3 && dx >= 0 && dx < 3) wv = wl[((dz * 3 + dy) * 3 + dx) * MF_CG + ch];
        const unsigned short hb = mf_bf16_bits(wv);
        vh[k] = (short)hb;
        vl[k] = (short)mf_bf16_bits(wv - mf_bf16_float(hb));
      }
      a_hi[dy][r] = vh;
      a_lo[dy][r] = vl;
    }
  float bv = bias ? bias[cg * MF_CG + ch] : 0.f;
  asm volatile("" : "+v"(bv));                       // the compiler's own load is awaited HERE, before any asm load is in flight

  // ---- units of this wave: u = 0..3 -> rows 4 ph .. 4 ph + 3 (this lane: row 4 ph + j), column pair u (output columns 2u, 2u + 1,
  //      input columns 2u .. 2u + 3 of the haloed image)
  int boff[4];                                        // halfword offset of B[k = 0][j] for dy = 0 (+ dy * EXP)
  bool pok[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    boff[u] = ch * MF_CS + (ph * 4 + j) * MF_EXP + 2 * u;
#pragma unroll
    for (int r = 0; r < 2; ++r) pok[u][r] = (y0 + ph * 4 + j) < g.H && (x0 + 2 * u + r) < g.W;
  }
  // read-back side of the output tile: thread t owns 16 bytes = position t >> 2 (row-major in the footprint), channel piece t & 3 --
  // a wave stores two full rows of the footprint, 512 contiguous bytes each.  (The first cut had every wave store its own 16
  // channels, 32 bytes per voxel: the output path then cost 180 of the kernel's 340 us -- profiles/r04_dwconv_mfma.txt.)
  const int op = tid >> 2, opiece = tid & 3;
  const int oy = op >> 3, ox = op & 7;
  const bool ook = (y0 + oy) < g.H && (x0 + ox) < g.W;
  const long obase = ((long)(y0 + oy) * g.W + (x0 + ox)) * C + opiece * 8;
  const int orb = oy * MF_RS + ox * MF_TS + opiece * 8;   // halfword offset of this thread's 16 bytes in the tile
  int ooff[4];                                        // halfword offset in the tile of this lane's (unit u, column 2u) value; column 2u + 1: + TS
#pragma unroll
  for (int u = 0; u < 4; ++u) ooff[u] = (ph * 4 + j) * MF_RS + (2 * u) * MF_TS + ch;
  auto flush = [&](int zo) {                          // tile of parity zo & 1 (written a step ago, a barrier in between) -> HBM
    const u32x4_t o = *reinterpret_cast<const u32x4_t*>(&otile[zo & 1][orb]);
    if (ook) *reinterpret_cast<u32x4_t*>(yn + (long)zo * plane_elems + obase) = o;
  };

  // ---- fused block: this wave's 16 positions of the footprint (two rows), lane (r, kb) = position 16 wave + r, channels kb*8 .. +7
  const int mr = lane & 15, mkb = lane >> 4;
  const int mp = wave * 16 + mr, my = mp >> 3, mxx = mp & 7;
  const bool mok = (y0 + my) < g.H && (x0 + mxx) < g.W;
  const long mbase = ((long)min(y0 + my, g.H - 1) * g.W + min(x0 + mxx, g.W - 1)) * C + mkb * 8;    // C = 32 (one channel group)
  const int mtile = my * MF_RS + mxx * MF_TS + mkb * 8;
  u32x4_t rres[MIX ? 3 : 1];                        // residual rows in flight: plane z travels in set (z - (zs - 1)) % 3
  auto issue_res = [&](int z, u32x4_t& rr) {          // R(z): this lane's 16 bytes of x at output plane z
    const unsigned short* ptr = xn + (long)z * plane_elems + mbase;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rr) : "v"(ptr) : "memory");
  };
  // channel mixer of output plane zo: tile of parity zo & 1 (written a step ago, a barrier in between) -> y.  `younger`: loads
  // requested after R(zo)
  auto mix = [&](int zo, u32x4_t& rr, int younger) {
    if constexpr (MIX) {
      const bf16x8_t bact = *reinterpret_cast<const bf16x8_t*>(&otile[zo & 1][mtile]);
      // biases from LDS (8 consecutive channels per lane): registers are what limits this kernel's occupancy
      f32x4_t acc2[2] = {*reinterpret_cast<const f32x4_t*>(&mbias[32 * MIXHC + mkb * 8]), *reinterpret_cast<const f32x4_t*>(&mbias[32 * MIXHC + mkb * 8 + 4])};
#pragma unroll
      for (int hc = 0; hc < MIXHC; ++hc) {
        f32x4_t acc1[2] = {*reinterpret_cast<const f32x4_t*>(&mbias[hc * 32 + mkb * 8]), *reinterpret_cast<const f32x4_t*>(&mbias[hc * 32 + mkb * 8 + 4])};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc1[mt] = Mma<bf16_t>::mma(w2s[(hc * 2 + mt) * 64 + lane], bact, acc1[mt]);
        float gg[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { gg[q] = acc1[0][q]; gg[4 + q] = acc1[1][q]; }
        const h8_t bhh = gelu_h8_from_f32(gg);
#pragma unroll
        for (int mo = 0; mo < 2; ++mo) acc2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3s[(mo * MIXHC + hc) * 64 + lane], bhh, acc2[mo], 0, 0, 0);
      }
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] = acc2[0][q]; v[4 + q] = acc2[1][q]; }
      if (mx.residual) {
        mf_wait_vm_upto<5>(younger);
        asm volatile("" : "+v"(rr) : : "memory");
        float rv[8];
        VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(&rr), rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += rv[q];
      }
      const long vo = (long)zo * plane_elems + mbase;
      if (mx.head_w) {
        // the block output rounded as the un-fused path stores it = the B fragment of logits^T[o][voxel] = sum_c head[o][c] y[c][voxel]
        const bf16x8_t ob = Mma<bf16_t>::from_floats(v);
        if (mx.store_y && mok) *reinterpret_cast<bf16x8_t*>(yn + vo) = ob;
        const f32x4_t hh = Mma<bf16_t>::mma(mx.head_w[lane], ob, f32x4_t{0.f, 0.f, 0.f, 0.f});
        if (mok) {
          float* hy = mx.head_y + ((long)n * g.D * g.H * g.W + (vo - mkb * 8) / C) * mx.n_head;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (mkb * 4 + i < mx.n_head) hy[mkb * 4 + i] = hh[i] + (mx.head_b ? mx.head_b[mkb * 4 + i] : 0.f);
        }
      } else if (mok) {
        VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(yn + vo), v);
      }
    }
  };

  f32x4_t acc[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 2; ++r) acc[u][r] = f32x4_t{bv, bv, bv, 0.f};
  float s1 = 0.f, s2 = 0.f;
  // PROBE 4: per-section cycle sums of this wave (wave-uniform values; s_memtime waits on lgkmcnt, i.e. also drains the wave's own
  // LDS operations at each stamp: the sections are slightly serialised against the untimed kernel, +5 % launch time measured)
  unsigned long long tk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, tstart = 0;
  auto stamp = [&](int k) {
    if constexpr (PROBE == 4) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tk[k] += now - tlast;
      tlast = now;
    }
  };
  if constexpr (PROBE == 4) { tstart = __builtin_amdgcn_s_memtime(); tlast = tstart; }

  // one z step: input plane gz is image[slot]; `ld` receives plane gz+PF, `cm` holds plane gz+1 and is committed after the compute
  // (fused block: `rn` receives the residual rows of plane gz, `ro` holds those of plane gz - 2)
  auto exists_p = [&](int z) { return z <= ze; };                       // plane load P(z) was / will be requested (z >= zs - 1 + PF)
  auto exists_r = [&](int z) { return MIX && mx.residual && z >= zs && z < ze; };
  auto step = [&](int gz, int slot, u32x4_t (&ld)[MF_CPT], u32x4_t (&cm)[MF_CPT], u32x4_t& rn, u32x4_t& ro) {
    if constexpr (MIX) {
      // requested after R(gz - 2): P(gz + 1), then R(gz - 1), P(gz + 2) (request order within a step: residual, then plane)
      if (gz - 2 >= zs) mix(gz - 2, ro, (exists_p(gz + 1) ? MF_CPT : 0) + (exists_r(gz - 1) ? 1 : 0) + (exists_p(gz + 2) ? MF_CPT : 0));
      stamp(8);
      if (exists_r(gz)) issue_res(gz, rn);
    } else {
      if (STORE && gz - 2 >= zs && PROBE != 3) flush(gz - 2);
    }
    if (gz + PF <= ze) issue(gz + PF, ld);
    stamp(0);
    // ---- the stencil of this plane: 3 row offsets x (2 output columns x hi/lo) instructions per unit
    const unsigned short* img = image[slot];
#pragma unroll
    for (int dy = 0; dy < ((PROBE == 1 || PROBE == 3) ? 0 : 3); ++dy) {
      s4_t bq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bq[u] = __builtin_bit_cast(s4_t, *reinterpret_cast<const u32x2a4_t*>(img + boff[u] + dy * MF_EXP));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[u][r] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a_hi[dy][r], bq[u], acc[u][r], 0, 0, 0);
      if (LO) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 2; ++r) acc[u][r] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a_lo[dy][r], bq[u], acc[u][r], 0, 0, 0);
      }
    }
    stamp(1);
    // ---- plane gz+1 into the other image slot
    if (gz + 1 <= ze) {
      // loads requested after plane gz+1: planes gz+2 .. gz+PF as far as the chunk goes, and (fused block) the residual rows of the
      // steps in between: R(gz - 1) and R(gz) at PF = 3
      landed(cm, MF_CPT * min(PF - 1, ze - gz - 1) + (exists_r(gz - 1) ? 1 : 0) + (exists_r(gz) ? 1 : 0));
      stamp(2);
      commit(slot ^ 1, cm, gz + 1);
      stamp(3);
    } else {
      // last step of the chunk: nothing is in flight any more.  The statement is here for csrc/asm_check.py: EVERY path through a step
      // executes a hand-written vmcnt wait after the step's load requests, so "no compiler instruction touches a staged register
      // before a hand-written wait" is a property of the listing that a static walk can establish (build-time check)
      mf_wait_vm<0>();
    }
    // ---- output plane gz-1 is complete: accumulator VGPR 0, lane = (channel, column) -> the workgroup's NDHWC tile of parity
    //      (gz-1) & 1; it leaves at the start of the next step, after this step's barrier
    if (gz - 1 >= zs && PROBE != 3) {
      unsigned short* ot = otile[STORE ? (gz - 1) & 1 : 0];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // the two columns of a unit round together (v_cvt_pk_bf16_f32)
        const bf2_t hb = __builtin_convertvector(f2_t{acc[u][0][0], acc[u][1][0]}, bf2_t);
        const unsigned int bits = __builtin_bit_cast(unsigned int, hb);
        if constexpr (STORE) {
          ot[ooff[u]] = (unsigned short)(bits & 0xffffu);
          ot[ooff[u] + MF_TS] = (unsigned short)(bits >> 16);
        }
        if constexpr (!MIX) {       // (the fused block takes no statistics: they come from the STORE = false pass)
          float r0 = __uint_as_float(bits << 16), r1 = __uint_as_float(bits & 0xffff0000u);
          if (!inner) { r0 = pok[u][0] ? r0 : 0.f; r1 = pok[u][1] ? r1 : 0.f; }
          s1 += r0; s2 = fmaf(r0, r0, s2);
          s1 += r1; s2 = fmaf(r1, r1, s2);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 2; ++r) acc[u][r] = f32x4_t{acc[u][r][1], acc[u][r][2], bv, 0.f};
    stamp(4);
    __syncthreads();
    stamp(5);
    if constexpr (PROBE == 4) tk[6] += 1;
  };

  // prologue: plane zs-1 -> image 0; planes zs (.. zs+1) already in flight.  Plane p travels in set (p - (zs-1)) % PF: step k
  // (gz = zs-1+k) requests plane gz+PF into set k % PF and commits plane gz+1 from set (k+1) % PF
#pragma unroll
  for (int q = 0; q < PF; ++q) issue(zs - 1 + q, stg[q]);
  landed(stg[0], MF_CPT * (PF - 1));
  commit(0, stg[0], zs - 1);
  __syncthreads();
  int slot = 0;
  for (int gz = zs - 1; gz <= ze; gz += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      if (gz + k <= ze) { step(gz + k, slot, stg[k], stg[(k + 1) % PF], rres[MIX ? k : 0], rres[MIX ? (k + 1) % 3 : 0]); slot ^= 1; }
    }
  }
  mf_wait_vm<0>();       // (asm_check.py: the epilogue below is reached through a hand-written wait on every static path)
  if constexpr (MIX) {
    // the last plane: its residual rows were requested at step ze - 1, nothing was requested after them.  Their register set is the
    // one of plane (ze - 1): (ze - 1 - (zs - 1)) % 3 -- a wave-uniform runtime index, resolved by a three-way branch
    const int rs = (ze - zs) % 3;
    if (rs == 0) mix(ze - 1, rres[0], 0);
    else if (rs == 1) mix(ze - 1, rres[MIX ? 1 : 0], 0);
    else mix(ze - 1, rres[MIX ? 2 : 0], 0);
  } else {
    if (STORE && PROBE != 3) flush(ze - 1);
  }

  if constexpr (PROBE == 4) {
    if constexpr (MIX) {
      if (mx.prof && lane == 0) {      // [N][slots][4 waves][9]: sections 0..7 as below, 8 = channel mixer (incl. its wait for the residual rows)
        tk[7] = __builtin_amdgcn_s_memtime() - tstart;
        float* o = mx.prof + (((long)n * g.slots + slot_id) * 4 + wave) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = (float)tk[k];
      }
    } else if (stats && lane == 0) {
      tk[7] = __builtin_amdgcn_s_memtime() - tstart;
      float* o = stats + (((long)n * g.slots + slot_id) * 2) * C + cg * MF_CG + wave * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (float)tk[k];
    }
    return;
  }
  if (!MIX && stats) {
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    if (j == 0) { red[wave][0][cl] = s1; red[wave][1][cl] = s2; }
    __syncthreads();
    if (tid < 2 * MF_CG) {
      const int which = tid / MF_CG, chn = tid % MF_CG;
      const float a = red[chn >> 4][which][chn & 15] + red[(chn >> 4) + 2][which][chn & 15];
      stats[(((long)n * g.slots + slot_id) * 2 + which) * C + cg * MF_CG + chn] = a;
    }
  }
}

void dwconv_mfma_launch(const void* x, void* y, const float* w, const float* bias, float* stats, const DwMarch& g, int variant,
                        hipStream_t s) {
  dim3 grid((unsigned)((long)g.slots * (g.C / MF_CG) * g.N)), block(256);
  const unsigned short* xp = (const unsigned short*)x;
  unsigned short* yp = (unsigned short*)y;
  const DwMix none{};
#define PYTC_MF(PFV, LOV, NM) hipLaunchKernelGGL((dwconv3d_k3_mfma_kernel<PFV, LOV, NM>), grid, block, 0, s, xp, yp, w, bias, stats, g, none)
#define PYTC_MF_STATS(PFV, LOV) hipLaunchKernelGGL((dwconv3d_k3_mfma_kernel<PFV, LOV, 0, false>), grid, block, 0, s, xp, yp, w, bias, stats, g, none)
  // variant: bit 0 = hi + lo weight instructions (16-bit weight mantissa; default: hi only = bf16 weights, what torch.autocast gives the
  // reference's Conv3d), bit 1 = two planes in flight instead of three.  Knob dwconv_mfma_probe (1 / 3; measurements only, WRONG results):
  // the kernel without its matrix instructions / without them and without the output path.
  const int probe = tuning_get("dwconv_mfma_probe", 0);
  if (!y) {                                           // statistics only (dw_entry has checked that statistics are requested)
    if (variant & 1) { if (variant & 2) PYTC_MF_STATS(2, true); else PYTC_MF_STATS(3, true); }
    else { if (variant & 2) PYTC_MF_STATS(2, false); else PYTC_MF_STATS(3, false); }
  }
  else if (probe == 1) PYTC_MF(3, false, 1);
  else if (probe == 3) PYTC_MF(3, false, 3);
  else if (probe == 4) PYTC_MF(3, false, 4);
  else if (variant & 1) { if (variant & 2) PYTC_MF(2, true, 0); else PYTC_MF(3, true, 0); }
  else { if (variant & 2) PYTC_MF(2, false, 0); else PYTC_MF(3, false, 0); }
#undef PYTC_MF
#undef PYTC_MF_STATS
}

// Fused block launch (see DwMix): C = 32, hidden width 64 / 96 / 128, bf16; variant bit 0 = hi + lo depthwise weights (must equal the
// variant of the statistics pass: the statistics are those of the tensor this kernel re-forms)
int dwmix_launch(const void* x, void* y, const float* w, const float* bias, const DwMarch& g, const DwMix& mx, int c_hid, int variant,
                 hipStream_t s) {
  dim3 grid((unsigned)((long)g.slots * g.N)), block(256);
  const unsigned short* xp = (const unsigned short*)x;
  unsigned short* yp = (unsigned short*)y;
#define PYTC_DM(LOV, HCV, PRB) hipLaunchKernelGGL((dwconv3d_k3_mfma_kernel<3, LOV, PRB, true, HCV>), grid, block, 0, s, xp, yp, w, bias, (float*)nullptr, g, mx)
  const bool lo = variant & 1;
  if (mx.prof) {                        // measurement only: section cycle counters (tools/r05_additivity.py phases_mix)
    if (c_hid != 64 || lo) return -1;
    PYTC_DM(false, 2, 4);
    return 0;
  }
  switch (c_hid) {
    case 64: if (lo) PYTC_DM(true, 2, 0); else PYTC_DM(false, 2, 0); break;
    case 96: if (lo) PYTC_DM(true, 3, 0); else PYTC_DM(false, 3, 0); break;
    case 128: if (lo) PYTC_DM(true, 4, 0); else PYTC_DM(false, 4, 0); break;
    default: return -1;
  }
#undef PYTC_DM
  return 0;
}

}  // namespace pytc
