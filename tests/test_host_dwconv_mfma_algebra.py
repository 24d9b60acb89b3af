"""Index algebra of the matrix-core depthwise conv (csrc/dwconv_mfma_kernels.hip), emulated in numpy on the CPU: the operand
assignment of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 products = 16 channels; pinned on hardware by
tools/probes/mfma4x4x4_layout_probe.hip) with
    A[b][i][k] = w_c[dz = 2 - i][dy = k - r][dx]          (row i = 3 and taps outside 0..2 are zero)
    B[b][k][j] = plane_c[y0 + k][x0 + j + dx]             (four consecutive rows of the haloed input plane)
    D[b][i][j] : VGPR i = 0 / 1 / 2 holds output plane gz-1 / gz / gz+1 at (row y0 + r, column x0 + j), in haloed coordinates
and the z-march's rotation of the accumulator tuple reproduce a 3x3x3 cross-correlation with zero padding.  This is the restatement
the kernel was written from; the GPU tests compare the kernel itself against an fp64 convolution."""
import numpy as np


def mfma_4x4x4_16b(A, B, C):
    """D[b] = A[b] @ B[b] + C[b] for the 16 blocks of one instruction (fp32 accumulation of exact products)."""
    return np.einsum("bik,bkj->bij", A, B) + C


def depthwise_via_mfma(x, w, bias):
    """x (D, H, W, 16) one 16-channel block group, w (3, 3, 3, 16), bias (16,) -> y (D, H, W, 16); H, W multiples of (2, 4)."""
    D, H, W, Cc = x.shape
    xp = np.zeros((D + 2, H + 2, W + 2, Cc), x.dtype)
    xp[1:-1, 1:-1, 1:-1] = x                                  # zero padding: the kernel zero-fills at the LDS commit
    y = np.zeros_like(x)
    A = np.zeros((3, 2, Cc, 4, 4), x.dtype)                   # [dx][r][block = channel][i][k]
    for dx in range(3):
        for r in range(2):
            for i in range(3):
                for k in range(4):
                    dy = k - r
                    if 0 <= dy < 3:
                        A[dx, r, :, i, k] = w[2 - i, dy, dx, :]
    for y0 in range(0, H, 2):                                 # unit: two output rows (r = 0, 1) sharing four input rows
        for x0 in range(0, W, 4):                             # ... and four output columns j
            acc = np.zeros((2, Cc, 4, 4), x.dtype)            # [r][block][VGPR i][lane j]
            acc[:, :, 0:3, :] = bias[None, :, None, None]
            for gz in range(-1, D + 1):                       # input planes -1 .. D (padded index gz + 1)
                for dx in range(3):
                    B = np.transpose(xp[gz + 1, y0:y0 + 4, x0 + dx:x0 + dx + 4, :], (2, 0, 1))      # [block][k][j]
                    for r in range(2):
                        acc[r] = mfma_4x4x4_16b(A[dx, r], B, acc[r])
                zo = gz - 1                                   # VGPR 0 is complete: output plane gz - 1
                if 0 <= zo < D:
                    for r in range(2):
                        y[zo, y0 + r, x0:x0 + 4, :] = acc[r][:, 0, :].T
                new = np.zeros_like(acc)                      # rotate: (v0, v1, v2, v3) <- (v1, v2, bias, 0)
                new[:, :, 0] = acc[:, :, 1]
                new[:, :, 1] = acc[:, :, 2]
                new[:, :, 2] = bias[None, :, None]
                acc = new
    return y


def reference(x, w, bias):
    D, H, W, Cc = x.shape
    xp = np.zeros((D + 2, H + 2, W + 2, Cc), x.dtype)
    xp[1:-1, 1:-1, 1:-1] = x
    y = np.zeros_like(x) + bias
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                y += xp[kz:kz + D, ky:ky + H, kx:kx + W, :] * w[kz, ky, kx, :]
    return y


def test_block_per_channel_mfma_formulation_is_the_3x3x3_cross_correlation():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 6, 8, 16))
    w = rng.standard_normal((3, 3, 3, 16))
    b = rng.standard_normal(16)
    np.testing.assert_allclose(depthwise_via_mfma(x, w, b), reference(x, w, b), rtol=1e-12, atol=1e-12)


def test_useful_fraction_of_the_operand():
    """9 of the 16 entries of an A block carry a tap (3 z taps x 3 y taps): 576 of the instruction's 1024 MACs are useful, and the
    stencil of (16 channels x 4 columns x 2 rows x 1 input plane) is 3 (dx) x 2 (r) = 6 instructions."""
    nz = 0
    for r in range(2):
        for i in range(3):
            for k in range(4):
                nz += 0 <= k - r < 3
    assert nz == 2 * 9 and 16 * 9 * 4 == 576
