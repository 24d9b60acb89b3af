"""GPU parity of the disk-backed volume reader's DEVICE half (csrc/volume_kernels.hip through inference/lazy_accessor.py) against
tests/golden/lazy_accessor.npz -- the REFERENCE's LazyVolumeAccessor (connectomics/inference/lazy.py:456-917) -- for every case and
storage back end, and against the numpy oracle of the same staged regions (oracle/accessor_oracle.py).

Tolerances: pure data movement (transpose, nearest resize, context / outer padding, binarise) and divide-K are exact; trilinear
resampling and the statistics modes carry fp32 rounding of weights and of the window mean / std (2e-6 of the value range)."""
import numpy as np
import pytest
import torch

from oracle import accessor_oracle as AO
from accessor_cases import CASES, _write_sources

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(CASES))
def test_device_accessor_matches_reference_fixture(name, golden_dir, tmp_path):
    from pytorch_connectomics_amd.inference.lazy_accessor import LazyVolumeAccessor
    g = np.load(golden_dir / "lazy_accessor.npz")
    vk, kw, reads, outer_mode, outer_val = CASES[name]
    exact = kw.get("kind") == "mask" or ("scale_factors" not in kw and kw.get("normalize_mode", "none") in ("none", "divide-255"))
    tol = dict(rtol=0, atol=0) if exact else dict(rtol=2e-5, atol=6e-4)
    for fmt, path in _write_sources(g, tmp_path, vk).items():
        with LazyVolumeAccessor(path, **kw) as acc:
            for i, (loc, size) in enumerate(reads):
                got = acc.read_patch(loc, size, outer_pad_mode=outer_mode, outer_pad_value=outer_val)
                want = g[f"{name}__patch{i}"]
                assert got.shape == want.shape and got.dtype == np.float32
                np.testing.assert_allclose(got, want, err_msg=f"{name} {fmt} patch {i}", **(tol if not exact else dict(rtol=1e-6, atol=1e-6)))
                # the kernel against the numpy execution of the SAME staged region
                np.testing.assert_allclose(got, AO.read_patch(acc, loc, size, outer_pad_mode=outer_mode, outer_pad_value=outer_val),
                                           rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(acc.load_full(), g[f"{name}__full"], **(tol if not exact else dict(rtol=1e-6, atol=1e-6)))
            if not acc.needs_window_statistics:
                lo, hi = (1, 0, 2), tuple(min(s, 9) for s in acc.padded_spatial_shape)
                reg = acc.read_region(lo, hi)
                np.testing.assert_allclose(reg, AO.finish(acc, AO.execute_staged(acc.stage_region(lo, hi))), rtol=1e-5, atol=1e-5)


def test_storage_dtypes_strides_and_window_normalize():
    """Every stored dtype through the resample kernel with a transposed, channel-last box; window_normalize against numpy for
    binarise / z-score / min-max / divide / percentile clip on a batch of windows with different statistics."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd.inference.lazy_accessor import _percentile_bounds
    rng = np.random.default_rng(2)
    for dt in ("uint8", "int8", "uint16", "int16", "uint32", "int32", "float32", "float64"):
        box = (rng.random((5, 6, 7, 2)) * 100).astype(dt)                       # stored (y, z, x, c): channel last, z/y swapped
        strides = (1, 7 * 2, 6 * 7 * 2, 2)                                      # (c, z, y, x) element strides
        nz, ny, nx = 6, 5, 7
        ident = lambda n: np.arange(n, dtype=np.int32)                          # noqa: E731
        i0 = torch.from_numpy(np.concatenate([ident(nz), ident(ny), ident(nx)])).cuda()
        f = torch.zeros(nz + ny + nx, dtype=torch.float32, device="cuda")
        raw = torch.from_numpy(box.reshape(-1).view(np.uint8).copy()).cuda()
        out = ops.resample_region(raw, dt, strides, 2, i0, i0.clone(), f, (nz, ny, nx)).cpu().numpy()
        np.testing.assert_array_equal(out, box.transpose(3, 1, 0, 2).astype(np.float32))
    with pytest.raises(TypeError, match="not supported"):
        ops.resample_region(raw, "float16", strides, 2, i0, i0, f, (nz, ny, nx))
    x = (torch.rand(5, 9, 10, 11, 2, generator=torch.Generator().manual_seed(3)) * torch.tensor([1.0, 3.0, 0.0, 250.0, 7.0]).view(5, 1, 1, 1, 1)
         + torch.tensor([0.0, -2.0, 4.0, 10.0, 0.5]).view(5, 1, 1, 1, 1))
    xn = x.numpy()
    for mode, ref in ((nat.NORM_ZSCORE, lambda w: (w - w.mean()) / w.std() if w.std() > 1e-8 else w),
                      (nat.NORM_MINMAX, lambda w: (w - w.min()) / (w.max() - w.min()) if w.max() > w.min() else w),
                      (nat.NORM_DIVIDE, lambda w: w / np.float32(255.0))):
        got = ops.window_normalize(x.cuda().clone(), mode=mode, divide=255.0).cpu().numpy()
        np.testing.assert_allclose(got, np.stack([ref(w) for w in xn]), rtol=2e-6, atol=2e-6)
    got = ops.window_normalize(x.cuda().clone(), binarize=True, threshold=2.5).cpu().numpy()
    np.testing.assert_array_equal(got, (xn > 2.5).astype(np.float32))
    flat = x.cuda().reshape(5, -1)
    clip = _percentile_bounds(flat, 0.05, 0.9)
    want_clip = np.stack([[np.percentile(w, 5), np.percentile(w, 90)] for w in xn]).astype(np.float32)
    np.testing.assert_allclose(clip.cpu().numpy(), want_clip, rtol=1e-6, atol=1e-6)
    got = ops.window_normalize(x.cuda().clone(), mode=nat.NORM_ZSCORE, clip=clip).cpu().numpy()
    np.testing.assert_allclose(got, np.stack([AO.smart_normalize(w, "normal", 0.05, 0.9) for w in xn]), rtol=1e-5, atol=1e-5)
    again = ops.window_normalize(x.cuda().clone(), mode=nat.NORM_ZSCORE, clip=clip).cpu().numpy()
    np.testing.assert_array_equal(got, again)                                    # fixed-order reductions
