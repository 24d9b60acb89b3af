// Stem (1x1x1 conv, 1 -> C channels) fused into the first depthwise 3x3x3 conv of the network (MedNeXt level 0).
//
// The stem's output s[v][c] = ws[c]*x[v] + bs[c] is the widest tensor of the network per byte of information (C channels
// derived from ONE input value per voxel); written and re-read it costs 2*C*2 B per voxel.  Here it is never formed:
//     t[v][c] = b1[c] + sum_tap w1[tap][c] * s[v+tap][c]            (s = 0 outside the volume: zero padding of the conv)
//             = (b1[c] + bs[c]*sum_tap w1[tap][c])  +  sum_tap (w1[tap][c]*ws[c]) * x[v+tap]  -  corr(v)[c]
// with corr(v)[c] = bs[c] * sum_{taps outside the volume} w1[tap][c], non-zero only for border voxels (wave-uniform skip
// in the interior).  One lane = 4 consecutive x voxels x 8 channels: the 9 x 6 neighbourhood of the 1-channel input from L1,
// the pre-multiplied taps from LDS once per group, 27*8*4 FMAs, four 16-byte stores (a voxel's 64-byte row by 4 lanes).  The GroupNorm
// statistics of the STORED (bf16-rounded) result are accumulated per lane over its voxels and reduced per workgroup in
// a fixed order through LDS -> per-workgroup slots (summed by pytc_groupnorm_finalize): deterministic, no atomics.
// The un-fused path rounds s to bf16 before the depthwise conv; this one keeps it in fp32 (closer to the fp32 reference).
#include "pytc_common.h"

namespace pytc {

constexpr int SD_C = 32;          // channels (MedNeXt base width)

struct StemDw { int D, H, W, GX; long groups; };   // GX = W / VX groups per x line, groups per sample

// wx [27][C] = w1[tap][c]*ws[c], wb [27][C] = w1[tap][c]*bs[c], cst [C] = b1[c] + sum_tap wb[tap][c] (host-side products),
// copied to LDS once per workgroup.  Lane = (group of 4 consecutive x voxels, 8-channel slice): the 8 x 27 taps of the
// slice are read from LDS once per GROUP (two ds_read_b128 per tap for 4 x 8 FMAs), the 9 x 6 neighbourhood of the
// 1-channel input comes from L1 as one 16-byte and two 4-byte loads per row.  (Earlier forms: one lane = one voxel x 32
// channels with the taps in LDS was LDS-bound at 860 us -- 216 tap reads per voxel; with the taps as scalar operands it
// waited on 54 s_load_dwordx16 per voxel, 781 us.)
// Without the compiler barrier at the top of the group loop hipcc keeps the lane's 216 taps (8 channels x 27) in VGPRs
// across the groups (256 VGPRs, 2 waves/SIMD, no LDS reads in the loop): 985 us against 436 us with the taps re-read from
// LDS per group (122 VGPRs, 4 waves/SIMD).  SD_IT = groups per lane: 2 -> 542 us, 4 -> 436, 8 -> 433, 16 -> 444.
constexpr int SD_IT = 4;
template <int SD_VX>      // consecutive x voxels per lane (4: 406 us at 8x112^3; 8 halves the tap reads but drops to 2 waves/SIMD: 561 us)
__global__ void __launch_bounds__(256)
stem_dwconv_k3_kernel(const float* __restrict__ x, const float* __restrict__ wx, const float* __restrict__ wb,
                      const float* __restrict__ cst, bf16_t* __restrict__ y, float* __restrict__ stats, StemDw g, int slots) {
  __shared__ __attribute__((aligned(16))) float lw[2 * 27 * SD_C + SD_C];   // wx | wb | cst
  __shared__ float red[16][256 + 4];
  for (int i = threadIdx.x; i < 27 * SD_C; i += 256) { lw[i] = wx[i]; lw[27 * SD_C + i] = wb[i]; }
  if (threadIdx.x < SD_C) lw[54 * SD_C + threadIdx.x] = cst[threadIdx.x];
  __syncthreads();
  const float* lwx = lw;
  const float* lwb = lw + 27 * SD_C;
  const float* lc = lw + 54 * SD_C;

  const int n = blockIdx.y;
  const int cg = threadIdx.x & 3, c0 = cg * 8;
  const long rps = (long)g.D * g.H * g.W;
  const float* xn = x + (long)n * rps;
  bf16_t* yn = y + (long)n * rps * SD_C;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  for (int it = 0; it < SD_IT; ++it) {
    asm volatile("" ::: "memory");                  // keep the tap reads of an iteration inside it
    const long G = ((long)blockIdx.x * SD_IT + it) * 64 + (threadIdx.x >> 2);
    const bool live = G < g.groups;
    const long Gc = live ? G : g.groups - 1;
    const int gx = (int)(Gc % g.GX);
    const long tq = Gc / g.GX;
    const int vy = (int)(tq % g.H), vz = (int)(tq / g.H);
    const int x0 = gx * SD_VX;
    float acc[SD_VX][8];
    {
      const float4 k0 = *reinterpret_cast<const float4*>(lc + c0), k1 = *reinterpret_cast<const float4*>(lc + c0 + 4);
#pragma unroll
      for (int v = 0; v < SD_VX; ++v) {
        acc[v][0] = k0.x; acc[v][1] = k0.y; acc[v][2] = k0.z; acc[v][3] = k0.w;
        acc[v][4] = k1.x; acc[v][5] = k1.y; acc[v][6] = k1.z; acc[v][7] = k1.w;
      }
    }
    const bool left_in = x0 > 0, right_in = x0 + SD_VX < g.W;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int dz = r / 3 - 1, dy = r % 3 - 1;
      const int z = vz + dz, yy = vy + dy;
      const bool row_in = z >= 0 && z < g.D && yy >= 0 && yy < g.H;
      const int zc = z < 0 ? 0 : (z >= g.D ? g.D - 1 : z), yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);
      const float* row = xn + ((long)zc * g.H + yc) * g.W + x0;
      const float lft = row[left_in ? -1 : 0], rgt = row[right_in ? SD_VX : SD_VX - 1];
      float in[SD_VX + 2];
      in[0] = (row_in && left_in) ? lft : 0.f;
#pragma unroll
      for (int q = 0; q < SD_VX / 4; ++q) {                               // W % VX == 0: 16-byte aligned
        const float4 mid = *reinterpret_cast<const float4*>(row + q * 4);
        in[1 + q * 4] = row_in ? mid.x : 0.f; in[2 + q * 4] = row_in ? mid.y : 0.f;
        in[3 + q * 4] = row_in ? mid.z : 0.f; in[4 + q * 4] = row_in ? mid.w : 0.f;
      }
      in[SD_VX + 1] = (row_in && right_in) ? rgt : 0.f;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int t = r * 3 + dx;
        const float4 w0 = *reinterpret_cast<const float4*>(lwx + t * SD_C + c0);
        const float4 w1 = *reinterpret_cast<const float4*>(lwx + t * SD_C + c0 + 4);
#pragma unroll
        for (int v = 0; v < SD_VX; ++v) {
          const float xv = in[v + dx];
          acc[v][0] = fmaf(w0.x, xv, acc[v][0]); acc[v][1] = fmaf(w0.y, xv, acc[v][1]);
          acc[v][2] = fmaf(w0.z, xv, acc[v][2]); acc[v][3] = fmaf(w0.w, xv, acc[v][3]);
          acc[v][4] = fmaf(w1.x, xv, acc[v][4]); acc[v][5] = fmaf(w1.y, xv, acc[v][5]);
          acc[v][6] = fmaf(w1.z, xv, acc[v][6]); acc[v][7] = fmaf(w1.w, xv, acc[v][7]);
        }
        // taps that fall outside the volume: the stem output there is 0, not bs -> take bs*w1 back out (borders only)
        const bool edge_l = dx == 0 && !left_in, edge_r = dx == 2 && !right_in;
        if (!row_in || edge_l || edge_r) {
          const float4 b0 = *reinterpret_cast<const float4*>(lwb + t * SD_C + c0);
          const float4 b1 = *reinterpret_cast<const float4*>(lwb + t * SD_C + c0 + 4);
#pragma unroll
          for (int v = 0; v < SD_VX; ++v) {
            const bool out = !row_in || (edge_l && v == 0) || (edge_r && v == SD_VX - 1);
            if (out) {
              acc[v][0] -= b0.x; acc[v][1] -= b0.y; acc[v][2] -= b0.z; acc[v][3] -= b0.w;
              acc[v][4] -= b1.x; acc[v][5] -= b1.y; acc[v][6] -= b1.z; acc[v][7] -= b1.w;
            }
          }
        }
      }
    }
    if (live) {
      const long vbase = ((long)vz * g.H + vy) * g.W + x0;
#pragma unroll
      for (int v = 0; v < SD_VX; ++v) {
        f32x8_t f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = acc[v][j];
        const bf16x8_t o = __builtin_convertvector(f, bf16x8_t);
        *reinterpret_cast<bf16x8_t*>(yn + (vbase + v) * SD_C + c0) = o;
        const f32x8_t q = __builtin_convertvector(o, f32x8_t);            // statistics of what is stored
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += q[j]; s2[j] = fmaf(q[j], q[j], s2[j]); }
      }
    }
  }

  // workgroup reduction in a fixed order: [16 values][256 lanes] scratch, 64 threads sum the 64 lanes of their channel slice
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[j][threadIdx.x] = s1[j]; red[8 + j][threadIdx.x] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < 2 * SD_C) {
    const int which = threadIdx.x / SD_C, c = threadIdx.x % SD_C;
    const float* col = red[which * 8 + (c & 7)];
    float a = 0.f;
    for (int l = 0; l < 64; ++l) a += col[l * 4 + (c >> 3)];
    stats[(((long)n * slots + blockIdx.x) * 2 + which) * SD_C + c] = a;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Matrix-core form.  The VALU kernel above spends 27 x 32 fp32 FMAs per voxel = 9.7 G lane-ops per 8 x 112^3 batch, 250 us
// of pure VALU issue on 256 CUs before any load, conversion or statistic: it is VALU-bound at ~0.4 ms while its 0.72 GB of
// output would take ~0.15 ms.  (An fp32-MFMA form was no faster: v_mfma_f32_16x16x4_f32 has the VALU's FLOP rate and its
// operand needed 28 gathers per lane.)  Here the conv is ONE f16 MFMA per 16 voxels x 16 channels:
//     t[c][v] = const[c] + sum_k A[c][k] * B[k][v],   A = f16(w1[k][c] * ws[c]) (32 x 27, zero-padded to K = 32),
//                                                       B = f16(x[v + tap k]) (0 outside the volume)
// with v_mfma_f32_16x16x32_f16 (fp32 accumulation).  f16 carries 11 mantissa bits for the 1-channel input and the fused taps
// -- 8x finer than the bf16 rounding the un-fused path applies to the stem output, and than the bf16 the result is stored in.
// A workgroup owns an 8 x 8 x 16 (z, y, x) block: its haloed 10 x 10 x 18 input block is converted to f16 into LDS once;
// a wave takes 16 consecutive x voxels at a time (N = voxel), lane (g = lane / 16, n = lane % 16) gathers its 8 K-values
// (taps 8g .. 8g+7 of voxel n) with eight ds_read_u16 (d16 / d16_hi: packed without VALU), two MFMAs cover the 32 channels
// with the A rows permuted so that the lane's 2 x 4 accumulators are EIGHT CONSECUTIVE channels (8g .. 8g+7) of voxel n:
// one 16-byte store per lane, four lanes per 64-byte voxel row -- the store pattern of the VALU kernel.
// Border blocks (wave-uniform) add the stem-bias term exactly: const = b1 and a second MFMA against the inside-indicator
// block, A_b = f16(w1[k][c] * bs[c]); interior blocks use const = b1 + sum_k w1*bs (cst).  Statistics as above.
constexpr int SM_XT = 16, SM_YT = 8, SM_ZT = 8;
constexpr int SM_RS = 20;                            // LDS row stride in halves: [pad, left, 16 interior (4-byte aligned), right, pad]
constexpr int SM_EY = SM_YT + 2, SM_EZ = SM_ZT + 2, SM_ROWS = SM_EY * SM_EZ, SM_HALO = SM_ROWS * SM_RS;
constexpr int SM_IMG_FRAG = 4 * 64;                  // h8 fragments: (wx | wb) x (MFMA 0 | 1) x lane
constexpr int SM_IMG_BYTES = SM_IMG_FRAG * 16 + 2 * SD_C * 4;

// zr = z extent of a workgroup (a multiple of SM_ZT): it walks its 8 x 16 (y, x) footprint through zr / 8 blocks.
// Everything a workgroup needs besides the input arrives ready-made in `image` (pytc_stem_dwconv3d_pack_mfma: the lanes' A
// fragments and the two constant vectors), so its set-up is six 16-byte loads per lane -- with the taps staged through LDS and
// the fragments built per workgroup the set-up was 85 us of a 254 us launch.
// Input staging: a haloed row is 16 aligned interior values (one float4 per lane-item) + two edge values, written to LDS as
// f16; 600 load items per block instead of 1800 scalar ones.
// Stores: the MFMA leaves the four 16-byte pieces of a voxel's 64-byte row in lanes 16 apart (lane = 16 * piece + voxel);
// they are first moved to lane 4 * voxel + piece (four ds_bpermute), so a wave's store is one contiguous 1 KB run.
struct StemMf { int D, H, W, bx, by, bz, zr; };      // blocks per axis (bz: z-chunks of zr planes)

// A fragments: lane holds A[m = lane % 16][k = 8 * (lane / 16) + j]; row m of MFMA q is channel 8 * (m / 4) + (m % 4) + 4 * q,
// so that the lane's two accumulators are the 8 consecutive channels 8 * (lane / 16) .. + 7 of voxel lane % 16
__global__ void __launch_bounds__(256)
stem_pack_mfma_kernel(const float* __restrict__ wx, const float* __restrict__ wb, const float* __restrict__ cst,
                      h8_t* __restrict__ frag, float* __restrict__ kc) {
  const int tid = threadIdx.x;
  const int mat = tid >> 7, q = (tid >> 6) & 1, lane = tid & 63;
  const int grp = lane >> 4, m = lane & 15;
  const int ch = 8 * (m >> 2) + (m & 3) + 4 * q;
  const float* w = mat ? wb : wx;
  h8_t v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * grp + j;
    v[j] = (_Float16)(k < 27 ? w[k * SD_C + ch] : 0.f);
  }
  frag[tid] = v;
  if (tid < SD_C) {
    float sb = 0.f;
    for (int k = 0; k < 27; ++k) sb += wb[k * SD_C + tid];
    kc[tid] = cst[tid];                              // interior blocks: b1 + sum_k w1 * bs
    kc[SD_C + tid] = cst[tid] - sb;                  // border blocks: b1 alone (the indicator MFMA adds the inside taps)
  }
}

template <bool STORE_PERMUTE>
__global__ void __launch_bounds__(256, 4)
stem_dwconv_k3_mfma_kernel(const float* __restrict__ x, const h8_t* __restrict__ frag, const float* __restrict__ kc,
                           bf16_t* __restrict__ y, float* __restrict__ stats, StemMf g, int slots) {
  // hx: f16 input block with halo (0 outside the volume) | hin: 1 inside the volume, 0 outside (border blocks)
  __shared__ __attribute__((aligned(16))) _Float16 hbuf[2 * SM_HALO];
  _Float16* const hx = hbuf;
  _Float16* const hin = hbuf + SM_HALO;
  __shared__ float red[4][2][SD_C];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, nn = lane & 15;
  const int c0 = 8 * grp;                            // the lane's 8 output channels
  const h8_t ax0 = frag[lane], ax1 = frag[64 + lane], ab0 = frag[128 + lane], ab1 = frag[192 + lane];
  const f32x4_t ki0 = *reinterpret_cast<const f32x4_t*>(kc + c0), ki1 = *reinterpret_cast<const f32x4_t*>(kc + c0 + 4);
  const f32x4_t kb0 = *reinterpret_cast<const f32x4_t*>(kc + SD_C + c0), kb1 = *reinterpret_cast<const f32x4_t*>(kc + SD_C + c0 + 4);

  const int n = blockIdx.y;
  int b = blockIdx.x;
  const int bxi = b % g.bx; b /= g.bx;
  const int byi = b % g.by;
  const int bzi = b / g.by;
  const int x0 = bxi * SM_XT, y0 = byi * SM_YT;
  const int zbeg = bzi * g.zr, zend = min(zbeg + g.zr, g.D);
  const bool left_in = x0 > 0, right_in = x0 + SM_XT < g.W;
  const bool xy_border = !left_in || !right_in || y0 == 0 || y0 + SM_YT >= g.H;
  const long rps = (long)g.D * g.H * g.W;
  const float* xn = x + (long)n * rps;
  bf16_t* yn = y + (long)n * rps * SD_C;

  // LDS address of the lane's tap j for tile 0 of its wave (row (0, wave) of the block); as address-space-3 pointers so that
  // the per-tile displacement folds into the ds_read offset field
  typedef const __attribute__((address_space(3))) _Float16* lds_h_ptr;
  lds_h_ptr tp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * grp + j;
    tp[j] = (lds_h_ptr)hbuf + (wave * SM_RS + 1 + nn + (k < 27 ? ((k / 9) * SM_EY + (k / 3) % 3) * SM_RS + k % 3 : 0));
  }
  // store side: this lane writes piece (lane % 4) of voxel (lane / 4), which the MFMA left in lane 16 * piece + voxel
  const int st_src = ((lane & 3) << 4) | (lane >> 2);
  const int st_vox = lane >> 2, st_piece = lane & 3;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  // ---- input staging.  Per block 100 haloed rows = 400 interior float4 items (2 per thread, second one for tid < 144) and 200
  // edge scalars (tid < 200).  The loads of block k+1 are issued BEFORE the tile loop of block k and converted after it: all
  // of a thread's loads are in flight together (written as one loop, hipcc waited for each load before issuing the next), and
  // their latency hides behind the tiles.
  constexpr int N_INT = SM_ROWS * 4, N_EDGE = SM_ROWS * 2;
  float4 pre_v[2];
  float pre_e;
  auto stage_issue = [&](int z0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = min(tid + 256 * u, N_INT - 1);
      const int r = it >> 2, p = it & 3;
      const int gy = y0 - 1 + r % SM_EY, gz = z0 - 1 + r / SM_EY;
      const int cy = min(max(gy, 0), g.H - 1), cz = min(max(gz, 0), g.D - 1);
      pre_v[u] = *reinterpret_cast<const float4*>(xn + ((long)cz * g.H + cy) * g.W + x0 + 4 * p);      // clamped row
    }
    {
      const int it = min(tid, N_EDGE - 1);
      const int r = it >> 1, right = it & 1;
      const int gy = y0 - 1 + r % SM_EY, gz = z0 - 1 + r / SM_EY;
      const int cy = min(max(gy, 0), g.H - 1), cz = min(max(gz, 0), g.D - 1);
      pre_e = xn[((long)cz * g.H + cy) * g.W + x0 + (right ? (right_in ? SM_XT : SM_XT - 1) : (left_in ? -1 : 0))];
    }
  };
  auto stage_commit = [&](int z0, bool border) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = tid + 256 * u;
      if (it < N_INT) {
        const int r = it >> 2, p = it & 3;
        const int gy = y0 - 1 + r % SM_EY, gz = z0 - 1 + r / SM_EY;
        const bool row_in = gy >= 0 && gy < g.H && gz >= 0 && gz < g.D;
        const float4 v = pre_v[u];
        h2_t* dst = reinterpret_cast<h2_t*>(&hx[r * SM_RS + 2 + 4 * p]);
        dst[0] = h2_t{(_Float16)(row_in ? v.x : 0.f), (_Float16)(row_in ? v.y : 0.f)};
        dst[1] = h2_t{(_Float16)(row_in ? v.z : 0.f), (_Float16)(row_in ? v.w : 0.f)};
        if (border) {
          const _Float16 one = (_Float16)(row_in ? 1.f : 0.f);
          h2_t* di = reinterpret_cast<h2_t*>(&hin[r * SM_RS + 2 + 4 * p]);
          di[0] = h2_t{one, one}; di[1] = h2_t{one, one};
        }
      }
    }
    if (tid < N_EDGE) {
      const int r = tid >> 1, right = tid & 1;
      const int gy = y0 - 1 + r % SM_EY, gz = z0 - 1 + r / SM_EY;
      const bool in = gy >= 0 && gy < g.H && gz >= 0 && gz < g.D && (right ? right_in : left_in);
      const int dsti = r * SM_RS + (right ? 2 + SM_XT : 1);
      hx[dsti] = (_Float16)(in ? pre_e : 0.f);
      if (border) hin[dsti] = (_Float16)(in ? 1.f : 0.f);
    }
  };

  stage_issue(zbeg);
  for (int z0 = zbeg; z0 < zend; z0 += SM_ZT) {
    const bool border = xy_border || z0 == 0 || z0 + SM_ZT >= g.D;        // workgroup-uniform
    __syncthreads();                                                       // the previous block's operand reads are done
    stage_commit(z0, border);
    __syncthreads();
    if (z0 + SM_ZT < zend) stage_issue(z0 + SM_ZT);                        // in flight during this block's tiles
    // 16 tiles per wave: tile (tz, h) of wave w is row (tz, 4 * h + w) of the block.  The two tiles of a plane share the lane's
    // tap addresses up to a compile-time displacement (the ds_read offset field); the addresses advance once per plane.  The
    // kernel is bound by instruction issue (a wave64 VALU instruction occupies its SIMD for 4 cycles; ~150 instructions per
    // tile were 170 us of issue on their own), so the tile body is kept minimal; the constants ride in as the MFMA's C operand.
    const f32x4_t k0 = border ? kb0 : ki0, k1 = border ? kb1 : ki1;
    bf16_t* ytile = yn + (((long)z0 * g.H + y0 + wave) * g.W + x0) * SD_C + (STORE_PERMUTE ? st_vox * SD_C + st_piece * 8 : nn * SD_C + c0);
#pragma unroll 1
    for (int tz = 0; tz < SM_ZT; ++tz) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int toff = 4 * h * SM_RS;                                    // compile-time
        h8_t bq;
#pragma unroll
        for (int j = 0; j < 8; ++j) bq[j] = tp[j][toff];
        f32x4_t a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax0, bq, k0, 0, 0, 0);
        f32x4_t a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax1, bq, k1, 0, 0, 0);
        if (border) {
          h8_t bi;
#pragma unroll
          for (int j = 0; j < 8; ++j) bi[j] = tp[j][SM_HALO + toff];
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ab0, bi, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ab1, bi, a1, 0, 0, 0);
        }
        f32x8_t f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[q] = a0[q]; f[4 + q] = a1[q]; }
        const bf16x8_t o = __builtin_convertvector(f, bf16x8_t);
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        i32x4_t sw = __builtin_bit_cast(i32x4_t, o);
        if constexpr (STORE_PERMUTE) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sw[q] = __shfl(sw[q], st_src, 64);
        }
        *reinterpret_cast<i32x4_t*>(ytile + (long)(4 * h) * g.W * SD_C) = sw;
        const f32x8_t qv = __builtin_convertvector(o, f32x8_t);           // statistics of what is stored, like every other kernel
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += qv[j]; s2[j] = fmaf(qv[j], qv[j], s2[j]); }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) tp[j] += SM_EY * SM_RS;
      ytile += (long)g.H * g.W * SD_C;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tp[j] -= SM_ZT * SM_EY * SM_RS;          // back to plane 0 for the next block
  }

  // fixed-order reduction: the 16 voxel lanes of a channel slice through xor-shuffles, the 4 waves through LDS
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
  }
  if (nn == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave][0][c0 + j] = s1[j]; red[wave][1][c0 + j] = s2[j]; }
  }
  __syncthreads();
  if (tid < 2 * SD_C) {
    const int which = tid / SD_C, c = tid % SD_C;
    stats[(((long)n * slots + blockIdx.x) * 2 + which) * SD_C + c] =
        ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
  }
}

}  // namespace pytc

using namespace pytc;

static int sd_vx(int) { return 4; }

static int sd_it() { return SD_IT; }

static bool sd_mfma(int D, int H, int W) {
  return W % SM_XT == 0 && H % SM_YT == 0 && D % SM_ZT == 0 && tuning_get("stem_mfma", 1) != 0;
}

// z extent per workgroup: the largest of 32 / 16 / 8 planes that still yields >= 256 workgroups for ONE sample.  A function of
// the volume shape only -- never of the batch size -- so the statistics partial-sum tree of a window does not depend on the
// batch it travels in (the engine's chunked == whole-volume bit-exactness relies on that).
static void sd_geom(pytc::StemMf& m, int D, int H, int W) {
  m.D = D; m.H = H; m.W = W;
  m.bx = W / SM_XT;
  m.by = (H + SM_YT - 1) / SM_YT;
  const int forced = tuning_get("stem_mfma_zr", 0);
  m.zr = SM_ZT;
  if (forced >= SM_ZT && forced % SM_ZT == 0) m.zr = forced;
  else
    for (int zr = 32; zr > SM_ZT; zr >>= 1)
      if ((long)m.bx * m.by * ((D + zr - 1) / zr) >= 256) { m.zr = zr; break; }
  m.bz = (D + m.zr - 1) / m.zr;
}

extern "C" int pytc_stem_dwconv3d_stat_slots(int D, int H, int W) {
  if (sd_mfma(D, H, W)) {
    StemMf m;
    sd_geom(m, D, H, W);
    return m.bx * m.by * m.bz;
  }
  const long groups = (long)D * H * (W / sd_vx(W));
  const long gpb = 64L * sd_it();       // voxel groups per workgroup (64 groups x 4 channel-group lanes = 256 lanes)
  return (int)((groups + gpb - 1) / gpb);
}

extern "C" int pytc_stem_dwconv3d_supported(int C_in, int C, int K) { return (C_in == 1 && C == SD_C && K == 3) ? 1 : 0; }

extern "C" int pytc_stem_dwconv3d_mfma_image_bytes(void) { return SM_IMG_BYTES; }

extern "C" int pytc_stem_dwconv3d_pack_mfma(const float* wx, const float* wb, const float* cst, void* image, void* stream) {
  PYTC_REQUIRE(wx && wb && cst && image, "stem_dwconv3d_pack_mfma: null pointer");
  hipLaunchKernelGGL(stem_pack_mfma_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, wx, wb, cst, (h8_t*)image,
                     (float*)((char*)image + SM_IMG_FRAG * 16));
  PYTC_LAUNCH_CHECK("stem_dwconv3d_pack_mfma");
  return PYTC_OK;
}

extern "C" int pytc_stem_dwconv3d_fwd(const float* x, const float* wx, const float* wb, const float* cst, const void* mfma_image,
                                      void* y, float* stats, int N, int D, int H, int W, int C, void* stream) {
  PYTC_REQUIRE(x && wx && wb && cst && y && stats, "stem_dwconv3d: null pointer");
  PYTC_REQUIRE(N >= 1 && D >= 1 && H >= 1 && W >= 4 && C == SD_C, "stem_dwconv3d: C must be 32");
  PYTC_REQUIRE(W % 4 == 0, "stem_dwconv3d: W must be a multiple of 4");
  const int slots = pytc_stem_dwconv3d_stat_slots(D, H, W);
  if (sd_mfma(D, H, W)) {
    PYTC_REQUIRE(mfma_image, "stem_dwconv3d: the f16-MFMA form needs the image of pytc_stem_dwconv3d_pack_mfma");
    StemMf m;
    sd_geom(m, D, H, W);
    const float* kc = (const float*)((const char*)mfma_image + SM_IMG_FRAG * 16);
    if (tuning_get("stem_store_permute", 1))
      hipLaunchKernelGGL(stem_dwconv_k3_mfma_kernel<true>, dim3(slots, N), dim3(256), 0, (hipStream_t)stream, x,
                         (const h8_t*)mfma_image, kc, (bf16_t*)y, stats, m, slots);
    else
      hipLaunchKernelGGL(stem_dwconv_k3_mfma_kernel<false>, dim3(slots, N), dim3(256), 0, (hipStream_t)stream, x,
                         (const h8_t*)mfma_image, kc, (bf16_t*)y, stats, m, slots);
    PYTC_LAUNCH_CHECK("stem_dwconv3d_mfma");
    return PYTC_OK;
  }
  const int vx = sd_vx(W);
  StemDw g{D, H, W, W / vx, (long)D * H * (W / vx)};
  hipLaunchKernelGGL(stem_dwconv_k3_kernel<4>, dim3(slots, N), dim3(256), 0, (hipStream_t)stream, x, wx, wb, cst, (bf16_t*)y,
                     stats, g, slots);
  PYTC_LAUNCH_CHECK("stem_dwconv3d");
  return PYTC_OK;
}
