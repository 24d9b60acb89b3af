"""Prototype (numpy, lane level) of a depthwise 3x3x3 convolution on the matrix cores in TOEPLITZ form -- the round-4 candidate of
DESIGN.md section 7.  Not product code: it pins down the index algebra a HIP kernel would implement and checks it against a direct
depthwise convolution, including the MFMA fragment layouts of `v_mfma_f32_16x16x32_bf16`:

    A fragment: lane l holds A[row = l & 15][k = (l >> 4) * 8 + i],  i = 0..7
    B fragment: lane l holds B[k = (l >> 4) * 8 + i][col = l & 15]
    D fragment: lane l holds D[row = (l >> 4) * 4 + r][col = l & 15], r = 0..3

Formulation, per channel c and output plane z of a footprint of 16 rows (y) x 14 columns (x):
    M = 16 output columns (14 used),  N = 16 output rows,  K = 32 = 2 tap groups x 16 input columns
    the 9 (dz, dy) tap pairs go two per MFMA -> 5 MFMAs (the tenth half is zero); tap group g of MFMA s is q = 2 s + g, (dz, dy) = divmod(q, 3)
    A[m][g * 16 + j] = w[c][dz][dy][j - m]            for 0 <= j - m <= 2 and m < 14, else 0     (a banded Toeplitz block per tap pair)
    B[g * 16 + j][n] = x[z + dz - 1][y0 + n + dy - 1][x0 - 1 + j][c]                         (zero outside the volume)
so that D[m][n] = out[z][y0 + n][x0 + m][c].  The input plane has to sit in LDS CHANNEL-MAJOR ([c][row][16 columns of the tile's window]):
lane (n, kg) then reads its 8 consecutive columns with one 16-byte LDS read; the NDHWC plane is transposed while it is staged, and the
result (a lane holds 4 consecutive x of ONE channel) is transposed back through LDS before it is stored.

    python tools/history/proto_toeplitz_dwconv.py        # self-check on a few shapes, prints the largest deviation from the direct convolution
"""
from __future__ import annotations

import numpy as np

TILE_Y, TILE_X, WIN = 16, 14, 16          # output rows / columns per tile; input columns per tile window (TILE_X + 2)


def bf16(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.asarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def direct(x: np.ndarray, w: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """reference: out[z,y,x,c] = bias[c] + sum_{dz,dy,dx} w[dz,dy,dx,c] * x[z+dz-1, y+dy-1, x+dx-1, c], zero padding, fp32"""
    D, H, W, C = x.shape
    xp = np.zeros((D + 2, H + 2, W + 2, C), np.float32)
    xp[1:-1, 1:-1, 1:-1] = x
    out = np.tile(bias.astype(np.float32), (D, H, W, 1))
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                out += w[dz, dy, dx] * xp[dz:dz + D, dy:dy + H, dx:dx + W]
    return out


def a_fragments(w_c: np.ndarray) -> np.ndarray:
    """the 5 A-operand fragments of one channel: (5, 64 lanes, 8) from its 3x3x3 taps w_c[dz][dy][dx]"""
    frag = np.zeros((5, 64, 8), np.float32)
    for s in range(5):
        for lane in range(64):
            m, kg = lane & 15, lane >> 4
            q = 2 * s + (kg >> 1)
            if q > 8 or m >= TILE_X:
                continue
            dz, dy = divmod(q, 3)
            for i in range(8):
                j = (kg & 1) * 8 + i
                if 0 <= j - m <= 2:
                    frag[s, lane, i] = w_c[dz, dy, j - m]
    return frag


def b_fragment(planes: np.ndarray, c: int, s: int) -> np.ndarray:
    """B-operand fragment of MFMA s for channel c: (64 lanes, 8).  planes[p][c][row 0..17][col 0..15] is the channel-major LDS image
    of input planes z-1, z, z+1 (p = dz); a lane's 8 values are CONSECUTIVE columns of one row: one 16-byte LDS read."""
    frag = np.zeros((64, 8), np.float32)
    for lane in range(64):
        n, kg = lane & 15, lane >> 4
        q = 2 * s + (kg >> 1)
        if q > 8:
            continue
        dz, dy = divmod(q, 3)
        j0 = (kg & 1) * 8
        frag[lane] = planes[dz, c, n + dy, j0:j0 + 8]
    return frag


def mfma_16x16x32(a: np.ndarray, b: np.ndarray, d: np.ndarray) -> np.ndarray:
    """D += A x B with the operands given as lane fragments (64, 8), (64, 8), (64, 4) -- the layouts at the top of this file"""
    A = np.zeros((16, 32), np.float32)
    B = np.zeros((32, 16), np.float32)
    for lane in range(64):
        A[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = a[lane]
        B[(lane >> 4) * 8:(lane >> 4) * 8 + 8, lane & 15] = b[lane]
    full = A.astype(np.float64) @ B.astype(np.float64)
    out = d.copy()
    for lane in range(64):
        for r in range(4):
            out[lane, r] += np.float32(full[(lane >> 4) * 4 + r, lane & 15])
    return out


def stage_plane(x: np.ndarray, z: int, y0: int, x0: int) -> np.ndarray:
    """NDHWC plane z -> channel-major LDS image [c][18 rows][16 columns] of the footprint's haloed window (zero outside the volume)"""
    D, H, W, C = x.shape
    img = np.zeros((C, TILE_Y + 2, WIN), np.float32)
    if 0 <= z < D:
        for row in range(TILE_Y + 2):
            yy = y0 - 1 + row
            if not 0 <= yy < H:
                continue
            for col in range(WIN):
                xx = x0 - 1 + col
                if 0 <= xx < W:
                    img[:, row, col] = x[z, yy, xx]
    return img


def toeplitz(x: np.ndarray, w: np.ndarray, bias: np.ndarray) -> np.ndarray:
    D, H, W, C = x.shape
    out = np.zeros((D, H, W, C), np.float32)
    afr = [a_fragments(w[..., c]) for c in range(C)]
    for y0 in range(0, H, TILE_Y):
        for x0 in range(0, W, TILE_X):
            ring = [stage_plane(x, -1, y0, x0), stage_plane(x, 0, y0, x0), stage_plane(x, 1, y0, x0)]       # planes z-1, z, z+1
            for z in range(D):
                planes = np.stack(ring)
                for c in range(C):
                    acc = np.zeros((64, 4), np.float32)
                    for s in range(5):
                        acc = mfma_16x16x32(afr[c][s], b_fragment(planes, c, s), acc)
                    for lane in range(64):                      # epilogue: lane holds 4 consecutive x of row n, channel c
                        n, mg = lane & 15, lane >> 4
                        for r in range(4):
                            m = mg * 4 + r
                            if m < TILE_X and y0 + n < H and x0 + m < W:
                                out[z, y0 + n, x0 + m, c] = acc[lane, r] + bias[c]
                ring = [ring[1], ring[2], stage_plane(x, z + 2, y0, x0)]
    return out


def main():
    rng = np.random.default_rng(0)
    worst = 0.0
    for shape in ((5, 16, 14, 4), (4, 20, 17, 3), (3, 7, 30, 2), (6, 33, 29, 2)):
        x = bf16(rng.standard_normal(shape))
        w = bf16(rng.standard_normal((3, 3, 3, shape[3])) * 0.3)
        b = rng.standard_normal(shape[3]).astype(np.float32)
        ref, got = direct(x, w, b), toeplitz(x, w, b)
        err = float(np.abs(ref - got).max())
        worst = max(worst, err)
        print(f"shape {shape}: max |toeplitz - direct| = {err:.3e}   (|out| up to {float(np.abs(ref).max()):.2f})")
    mfma_per_output = 5 / (TILE_X * TILE_Y)
    print(f"MFMAs per output voxel-channel: {mfma_per_output:.4f}  (16 cycles each: {16 * mfma_per_output:.3f} cycles; the packed-f16 VALU form: ~1.03)")
    assert worst < 5e-5, worst
    print("ok")


if __name__ == "__main__":
    main()
