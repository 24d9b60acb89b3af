"""Index algebra of the two resampling depthwise kernels of round 4, restated on the CPU and checked against torch:
  * csrc/dwconvT_tile_kernels.hip -- the transposed conv (k 3, stride 2, pad 1, output_padding 1) as 2x2x2 output CELLS per input voxel:
    per axis an even output position p = 2m reads (input m-1, tap 2) and (input m, tap 0), an odd one p = 2m+1 reads (input m, tap 1);
    with this package's placement the output grid is the padded one: position 0 of every axis is a zero face (hip_ops.dwconv3d);
  * csrc/dwconv_s2_kernels.hip -- the stride-2 conv as a z-march over a ring of three LDS slots: input plane iz lives in slot (iz+1) % 3,
    output plane zo reads planes 2zo-1, 2zo, 2zo+1, and the two planes requested for the next output overwrite the two oldest slots.
The GPU tests compare the kernels bit for bit with the kernels they replace; this file pins the algebra those kernels share."""
import numpy as np
import torch
import torch.nn.functional as F


def transposed_by_cells(x, w, bias):
    """x (D, H, W, C), w (3, 3, 3, C) tap-major as the kernels take it, bias (C,) -> y (2D, 2H, 2W, C) on the padded output grid."""
    D, H, W, C = x.shape
    y = np.zeros((2 * D, 2 * H, 2 * W, C))
    for mz in range(D):
        for my in range(H):
            for mx in range(W):
                xin = np.zeros((2, 2, 2, C))
                for a in range(2):
                    for b in range(2):
                        for d in range(2):
                            xin[a, b, d] = x[max(mz - 1 + a, 0), max(my - 1 + b, 0), max(mx - 1 + d, 0)]
                for pz in range(2):
                    for py in range(2):
                        for px in range(2):
                            acc = bias.copy()
                            for a in range(pz, 2):
                                for b in range(py, 2):
                                    for d in range(px, 2):
                                        kz, ky, kx = (1 if pz else (0 if a else 2)), (1 if py else (0 if b else 2)), (1 if px else (0 if d else 2))
                                        acc = acc + xin[a, b, d] * w[kz, ky, kx]
                            P = (2 * mz + pz, 2 * my + py, 2 * mx + px)
                            y[P] = 0.0 if 0 in P else acc
    return y


def test_cell_form_of_the_transposed_depthwise_conv_matches_torch():
    rng = np.random.default_rng(1)
    D, H, W, C = 3, 4, 5, 6
    x, w, b = rng.standard_normal((D, H, W, C)), rng.standard_normal((3, 3, 3, C)), rng.standard_normal(C)
    got = transposed_by_cells(x, w, b)
    # ConvTranspose3d(k 3, s 2, p 1, output_padding 1): out[o] += in[m] * W[k] with o = 2m - 1 + k  <=>  padded position p = o + 1
    wt = torch.from_numpy(w).permute(3, 0, 1, 2).unsqueeze(1)                       # (C, 1, 3, 3, 3)
    ref = F.conv_transpose3d(torch.from_numpy(x).permute(3, 0, 1, 2).unsqueeze(0), wt, torch.from_numpy(b), stride=2, padding=1,
                             output_padding=1, groups=C)[0].permute(1, 2, 3, 0).numpy()   # positions o = 0 .. 2D-1
    # this package's grid is shifted by one: p = o + 1, the face p = 0 is zero and the reference's last position o = 2D-1 falls off
    np.testing.assert_allclose(got[1:, 1:, 1:], ref[:-1, :-1, :-1], rtol=1e-12, atol=1e-12)
    assert not got[0].any() and not got[:, 0].any() and not got[:, :, 0].any()


def test_stride2_ring_march_matches_torch():
    rng = np.random.default_rng(2)
    D, H, W, C = 9, 7, 8, 4
    x, w, b = rng.standard_normal((D, H, W, C)), rng.standard_normal((3, 3, 3, C)), rng.standard_normal(C)
    Do, Ho, Wo = (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1

    def plane(iz):                                   # haloed, zero-filled input plane as `deposit` writes it
        p = np.zeros((2 * Ho + 1, 2 * Wo + 1, C))
        if 0 <= iz < D:
            for ly in range(2 * Ho + 1):
                for lx in range(2 * Wo + 1):
                    iy, ix = ly - 1, lx - 1
                    if 0 <= iy < H and 0 <= ix < W:
                        p[ly, lx] = x[iz, iy, ix]
        return p

    ring = [None, None, None]
    for iz in (-1, 0, 1):
        ring[(iz + 1) % 3] = plane(iz)
    y = np.zeros((Do, Ho, Wo, C))
    for zo in range(Do):
        planes = [ring[(2 * zo) % 3], ring[(2 * zo + 1) % 3], ring[(2 * zo + 2) % 3]]      # input planes 2zo-1, 2zo, 2zo+1
        for vy in range(Ho):
            for vx in range(Wo):
                acc = b.copy()
                for kz in range(3):
                    for ky in range(3):
                        for kx in range(3):
                            acc = acc + planes[kz][2 * vy + ky, 2 * vx + kx] * w[kz, ky, kx]
                y[zo, vy, vx] = acc
        if zo + 1 < Do:                                   # the next output's two new planes replace the two oldest
            for iz in (2 * zo + 2, 2 * zo + 3):
                ring[(iz + 1) % 3] = plane(iz)
    wt = torch.from_numpy(w).permute(3, 0, 1, 2).unsqueeze(1)
    ref = F.conv3d(torch.from_numpy(x).permute(3, 0, 1, 2).unsqueeze(0), wt, torch.from_numpy(b), stride=2, padding=1, groups=C)[0]
    np.testing.assert_allclose(y, ref.permute(1, 2, 3, 0).numpy(), rtol=1e-12, atol=1e-12)
