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
template <int PF, bool LO, int PROBE = 0, bool STORE = true>
__global__ void __launch_bounds__(256, 4)
dwconv3d_k3_mfma_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, const float* __restrict__ w,
                        const float* __restrict__ bias, float* __restrict__ stats, DwMarch g) {
  static_assert(PF == 2 || PF == 3, "two or three planes in flight");
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
  auto landed = [&](u32x4_t (&st)[MF_CPT], int younger) {
    if (younger <= 0) mf_wait_vm<0>();
    else if (younger == 1) mf_wait_vm<MF_CPT>();
    else mf_wait_vm<2 * MF_CPT>();
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

  f32x4_t acc[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 2; ++r) acc[u][r] = f32x4_t{bv, bv, bv, 0.f};
  float s1 = 0.f, s2 = 0.f;
  // PROBE 4: per-section cycle sums of this wave (wave-uniform values; s_memtime waits on lgkmcnt, i.e. also drains the wave's own
  // LDS operations at each stamp: the sections are slightly serialised against the untimed kernel, +5 % launch time measured)
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, tstart = 0;
  auto stamp = [&](int k) {
    if constexpr (PROBE == 4) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tk[k] += now - tlast;
      tlast = now;
    }
  };
  if constexpr (PROBE == 4) { tstart = __builtin_amdgcn_s_memtime(); tlast = tstart; }

  // one z step: input plane gz is image[slot]; `ld` receives plane gz+PF, `cm` holds plane gz+1 and is committed after the compute
  auto step = [&](int gz, int slot, u32x4_t (&ld)[MF_CPT], u32x4_t (&cm)[MF_CPT]) {
    if (STORE && gz - 2 >= zs && PROBE != 3) flush(gz - 2);
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
      landed(cm, min(PF - 1, ze - gz - 1));           // planes requested after plane gz+1 (gz+2 .. gz+PF, as far as the chunk goes)
      stamp(2);
      commit(slot ^ 1, cm, gz + 1);
      stamp(3);
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
        float r0 = __uint_as_float(bits << 16), r1 = __uint_as_float(bits & 0xffff0000u);
        if (!inner) { r0 = pok[u][0] ? r0 : 0.f; r1 = pok[u][1] ? r1 : 0.f; }
        s1 += r0; s2 = fmaf(r0, r0, s2);
        s1 += r1; s2 = fmaf(r1, r1, s2);
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
  landed(stg[0], PF - 1);
  commit(0, stg[0], zs - 1);
  __syncthreads();
  int slot = 0;
  for (int gz = zs - 1; gz <= ze; gz += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      if (gz + k <= ze) { step(gz + k, slot, stg[k], stg[(k + 1) % PF]); slot ^= 1; }
    }
  }
  if (STORE && PROBE != 3) flush(ze - 1);

  if constexpr (PROBE == 4) {
    if (stats && lane == 0) {
      tk[7] = __builtin_amdgcn_s_memtime() - tstart;
      float* o = stats + (((long)n * g.slots + slot_id) * 2) * C + cg * MF_CG + wave * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (float)tk[k];
    }
    return;
  }
  if (stats) {
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
#define PYTC_MF(PFV, LOV, NM) hipLaunchKernelGGL((dwconv3d_k3_mfma_kernel<PFV, LOV, NM>), grid, block, 0, s, xp, yp, w, bias, stats, g)
#define PYTC_MF_STATS(PFV, LOV) hipLaunchKernelGGL((dwconv3d_k3_mfma_kernel<PFV, LOV, 0, false>), grid, block, 0, s, xp, yp, w, bias, stats, g)
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

}  // namespace pytc
