"""Index algebra of the matrix-core depthwise conv (csrc/dwconv_mfma_kernels.hip), emulated in numpy on the CPU: the operand
assignment of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 products = 16 channels; pinned on hardware by
tools/probes/mfma4x4x4_layout_probe.hip) with
    A[b][i][k] = w_c[dz = 2 - i][dy][dx = k - r]          (row i = 3 and taps outside 0..2 are zero)
    B[b][k][j] = plane_c[y0 + j + dy][x0 + k]             (four consecutive COLUMNS of one row of the haloed input plane: round 5,
                                                           the LDS image keeps x innermost -- see tools/lds_conflict_model.py)
    D[b][i][j] : VGPR i = 0 / 1 / 2 holds output plane gz-1 / gz / gz+1 at (row y0 + j, column x0 + r), in haloed coordinates
and the z-march's rotation of the accumulator tuple reproduce a 3x3x3 cross-correlation with zero padding.  This is the restatement
the kernel was written from; the GPU tests compare the kernel itself against an fp64 convolution."""
import numpy as np


def mfma_4x4x4_16b(A, B, C):
    """D[b] = A[b] @ B[b] + C[b] for the 16 blocks of one instruction (fp32 accumulation of exact products)."""
    return np.einsum("bik,bkj->bij", A, B) + C


def depthwise_via_mfma(x, w, bias):
    """x (D, H, W, 16) one 16-channel block group, w (3, 3, 3, 16), bias (16,) -> y (D, H, W, 16); H, W multiples of (4, 2)."""
    D, H, W, Cc = x.shape
    xp = np.zeros((D + 2, H + 2, W + 2, Cc), x.dtype)
    xp[1:-1, 1:-1, 1:-1] = x                                  # zero padding: the kernel zero-fills at the LDS commit
    y = np.zeros_like(x)
    A = np.zeros((3, 2, Cc, 4, 4), x.dtype)                   # [dy][r][block = channel][i][k]
    for dy in range(3):
        for r in range(2):
            for i in range(3):
                for k in range(4):
                    dx = k - r
                    if 0 <= dx < 3:
                        A[dy, r, :, i, k] = w[2 - i, dy, dx, :]
    for y0 in range(0, H, 4):                                 # unit: four output rows j
        for x0 in range(0, W, 2):                             # ... and two output columns (r = 0, 1) sharing four input columns
            acc = np.zeros((2, Cc, 4, 4), x.dtype)            # [r][block][VGPR i][lane j]
            acc[:, :, 0:3, :] = bias[None, :, None, None]
            for gz in range(-1, D + 1):                       # input planes -1 .. D (padded index gz + 1)
                for dy in range(3):
                    B = np.transpose(xp[gz + 1, y0 + dy:y0 + dy + 4, x0:x0 + 4, :], (2, 1, 0))      # [block][k = column][j = row]
                    for r in range(2):
                        acc[r] = mfma_4x4x4_16b(A[dy, r], B, acc[r])
                zo = gz - 1                                   # VGPR 0 is complete: output plane gz - 1
                if 0 <= zo < D:
                    for r in range(2):
                        y[zo, y0:y0 + 4, x0 + r, :] = acc[r][:, 0, :].T
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
    x = rng.standard_normal((5, 8, 6, 16))
    w = rng.standard_normal((3, 3, 3, 16))
    b = rng.standard_normal(16)
    np.testing.assert_allclose(depthwise_via_mfma(x, w, b), reference(x, w, b), rtol=1e-12, atol=1e-12)


def test_useful_fraction_of_the_operand():
    """9 of the 16 entries of an A block carry a tap (3 z taps x 3 x taps): 576 of the instruction's 1024 MACs are useful, and the
    stencil of (16 channels x 4 rows x 2 columns x 1 input plane) is 3 (dy) x 2 (r) = 6 instructions."""
    nz = 0
    for r in range(2):
        for i in range(3):
            for k in range(4):
                nz += 0 <= k - r < 3
    assert nz == 2 * 9 and 16 * 9 * 4 == 576


def test_lds_layout_of_the_kernel_is_conflict_free_in_the_bank_model():
    """tools/lds_conflict_model.py restates the LDS addresses of every DS instruction of a plane step (commit writes, operand reads,
    tile writes, flush reads) and the bank rules of MI355X_MICROARCH.md.  The round-4 layout (y innermost, 120-halfword channel
    stride) costs 848 LDS-array cycles per workgroup step -- rocprofv3 measured SQ_LDS_BANK_CONFLICT = 60 % of SQ_LDS_IDX_ACTIVE and a
    step of 3 339 cycles at 4 workgroups per CU (profiles/r05_additivity.txt) --, the round-5 layout (x innermost, channel stride 81
    dwords, rows 8 dwords apart, tile rows padded by 4 dwords: the constants of csrc/dwconv_mfma_kernels.hip) 376, i.e. every
    instruction at its conflict-free minimum."""
    import importlib.util
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("lds_conflict_model", root / "tools" / "lds_conflict_model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    src = (root / "pytorch_connectomics_amd" / "csrc" / "dwconv_mfma_kernels.hip").read_text()
    exp = int(re.search(r"constexpr int MF_EXP = (\d+);", src).group(1))
    assert re.search(r"constexpr int MF_CS = MF_EY \* MF_EXP \+ 2;", src) and re.search(r"constexpr int MF_RS = MF_TX \* MF_TS \+ 8;", src)
    cs, rs = 10 * exp + 2, 8 * 32 + 8
    old = m.model(120, 12, 32)
    new = m.model_x(cs, exp, rs, 32)
    assert old["total"] == 848 and old["commit_b16"][0] == 416
    assert new["total"] == 376
    assert new["bread"][0] == 4 * new["bread"][1] and new["tile_w16"][0] == 2 * new["tile_w16"][1] and new["commit_b16"][0] <= 2 * new["commit_b16"][1]
