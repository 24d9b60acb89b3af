"""Lazy sliding-window inference with per-window test-time augmentation and a mask volume (reference inference/lazy.py:986-1258:
its loop hands every window batch to a TTAPredictor) -- the host orchestration on the CPU.

The product has no CPU path, so the kernel module (`hip_ops`) is replaced by the small torch stand-ins below (test infrastructure,
same idea as tests/test_host_distributed_inference.py): window gather with the reader's outer padding, blending, activations, the
ensemble update.  Everything above the kernels -- which views run, inverse views, activation-before-ensemble order, per-channel
ensemble modes, mask application, channel selection, blending of the ensembled windows -- is the product code, checked against
tests/golden/lazy_tta.npz, the output of the REFERENCE's own lazy loop on the same volume / mask / network.  The same fixtures
meet the real kernels in tests/test_gpu_lazy_chunked.py."""
import numpy as np
import pytest
import torch

from lazy_tta_cases import LAZY_TTA_CASES, lazy_tta_cfg
from test_host_distributed_inference import _CpuOps


class _Ops(_CpuOps):
    @staticmethod
    def _gather_plain(vol, starts, roi, pad_mode="constant", cval=0.0, **_kw):
        """(C,Z,Y,X) -> (B, *roi, C); a window overhanging the box is padded like np.pad (constant / reflect / replicate)."""
        mode = {"constant": "constant", "reflect": "reflect", "replicate": "edge", "edge": "edge"}[str(pad_mode)]
        ext = vol.shape[1:]
        out = []
        for s in starts:
            lo = [max(0, s[a]) for a in range(3)]
            hi = [min(ext[a], s[a] + roi[a]) for a in range(3)]
            inner = vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].numpy()
            pads = [(0, 0)] + [(lo[a] - s[a], s[a] + roi[a] - hi[a]) for a in range(3)]
            kw = dict(constant_values=float(cval)) if mode == "constant" else {}
            out.append(torch.from_numpy(np.pad(inner, pads, mode=mode, **kw)).permute(1, 2, 3, 0))
        return torch.stack(out).contiguous()

    # ---- TTA view codes, affinity channel maps (csrc/window_kernels.hip: view_src, blend_accumulate[_mapped], blend_weight_shifted) ----
    @staticmethod
    def _to_view(win, view):
        """canonical window (..., z, y, x, C) -> the view the network sees: out[z, y, x] = win[T(F(z, y, x))]."""
        from pytorch_connectomics_amd import _native as nat
        if view & nat.VIEW_SWAP_YX:
            win = win.transpose(-3, -2)
        dims = [d for d, bit in ((-4, nat.VIEW_FLIP_Z), (-3, nat.VIEW_FLIP_Y), (-2, nat.VIEW_FLIP_X)) if view & bit]
        return torch.flip(win, dims) if dims else win

    @staticmethod
    def _from_view(pred, view):
        """prediction of a view (..., z, y, x, C) -> canonical window frame (the inverse of `_to_view`)."""
        from pytorch_connectomics_amd import _native as nat
        dims = [d for d, bit in ((-4, nat.VIEW_FLIP_Z), (-3, nat.VIEW_FLIP_Y), (-2, nat.VIEW_FLIP_X)) if view & bit]
        pred = torch.flip(pred, dims) if dims else pred
        return pred.transpose(-3, -2) if view & nat.VIEW_SWAP_YX else pred

    @staticmethod
    def _window_map(wz, wy, wx, combine, floor_w, border):
        from pytorch_connectomics_amd.inference.window import _combine_axes
        w = _combine_axes([wz, wy, wx], combine, floor_w, "cpu", torch.float32).clone()
        if border is not None and any(int(b) for b in border):
            keep = torch.zeros_like(w)
            bz, by, bx = (int(b) for b in border)
            keep[bz:w.shape[0] - bz, by:w.shape[1] - by, bx:w.shape[2] - bx] = 1.0
            w = w * keep
        return w

    @staticmethod
    def _land(dst, src, start, lo=(0, 0, 0)):
        """dst[start + lo ...] += src, clipped to dst (voxels of a window outside the accumulator are skipped)."""
        ext, size = dst.shape[-3:], src.shape[-3:]
        a = [start[i] + lo[i] for i in range(3)]
        l = [max(0, a[i]) for i in range(3)]
        h = [min(ext[i], a[i] + size[i]) for i in range(3)]
        if any(h[i] <= l[i] for i in range(3)):
            return
        d = tuple(slice(l[i], h[i]) for i in range(3))
        s_ = tuple(slice(l[i] - a[i], h[i] - a[i]) for i in range(3))
        dst[(Ellipsis,) + d] += src[(Ellipsis,) + s_]

    @classmethod
    def gather_windows(cls, vol, starts, roi, *, view=0, pad_mode="constant", cval=0.0, **_kw):  # noqa: F811  (view-aware form)
        return cls._to_view(cls._gather_plain(vol, starts, roi, pad_mode=pad_mode, cval=cval), view).contiguous()

    @classmethod
    def blend_accumulate(cls, pred, starts, value, weight, wz, wy, wx, *, view=0, combine=0, floor_w=1e-5, border=None):
        assert len(starts) == pred.shape[0]
        w = cls._window_map(wz, wy, wx, combine, floor_w, border)
        canon = cls._from_view(pred.float(), view)
        for i, s in enumerate(starts):
            cls._land(value, canon[i].permute(3, 0, 1, 2) * w, s)
            if weight is not None:
                cls._land(weight, w, s)

    @classmethod
    def blend_accumulate_mapped(cls, pred, starts, value, weight, wz, wy, wx, chan_src, chan_shift, *, view=0, combine=0, floor_w=1e-5,
                                border=None):
        """output channel d <- canonical prediction channel chan_src[d] displaced by chan_shift[d]: the value predicted at q lands
        at p = q + shift, weighted by the window map at p; p outside the window is dropped."""
        assert len(starts) == pred.shape[0]
        w = cls._window_map(wz, wy, wx, combine, floor_w, border)
        canon = cls._from_view(pred.float(), view)
        roi = canon.shape[1:4]
        for i, s in enumerate(starts):
            for d, (src, sh) in enumerate(zip(chan_src, chan_shift)):
                q_lo = [max(0, -int(sh[a])) for a in range(3)]
                q_hi = [min(roi[a], roi[a] - int(sh[a])) for a in range(3)]
                if any(q_hi[a] <= q_lo[a] for a in range(3)):
                    continue
                q = tuple(slice(q_lo[a], q_hi[a]) for a in range(3))
                p_lo = [q_lo[a] + int(sh[a]) for a in range(3)]
                pbox = tuple(slice(p_lo[a], p_lo[a] + q_hi[a] - q_lo[a]) for a in range(3))
                cls._land(value[d], canon[i][q + (int(src),)] * w[pbox], s, p_lo)
            if weight is not None:
                cls._land(weight, w, s)

    @classmethod
    def blend_weight_shifted(cls, starts, roi, weight, wz, wy, wx, shift, *, combine=0, floor_w=1e-5, border=None):
        """weight += the window map over the positions p of each window whose source p - shift lies inside the window."""
        w = cls._window_map(wz, wy, wx, combine, floor_w, border)
        p_lo = [max(0, int(shift[a])) for a in range(3)]
        p_hi = [min(int(roi[a]), int(roi[a]) + int(shift[a])) for a in range(3)]
        if any(p_hi[a] <= p_lo[a] for a in range(3)):
            return
        box = tuple(slice(p_lo[a], p_hi[a]) for a in range(3))
        for s in starts:
            cls._land(weight, w[box], s, p_lo)

    @staticmethod
    def normalize_covered(value, weight):
        value.copy_(torch.where(weight > 0, value / torch.where(weight > 0, weight, torch.ones_like(weight)), torch.zeros_like(value)))

    @staticmethod
    def channel_activation(value, c0, c1, act, scale=1.0, *, channels_last=False):
        from pytorch_connectomics_amd import _native as nat
        v = value[..., c0:c1] if channels_last else value[c0:c1]
        if act == nat.ACT_SIGMOID:
            v.copy_(torch.sigmoid(scale * v))
        elif act == nat.ACT_TANH:
            v.copy_(torch.tanh(scale * v))
        elif act == nat.ACT_SOFTMAX:
            v.copy_(torch.softmax(v, dim=-1 if channels_last else 0))
        else:
            raise AssertionError(act)


    @staticmethod
    def ensemble_update_masked(stat, count, x, cover, mode):
        """validity-aware streaming ensemble (csrc/window_kernels.hip pytc_ensemble_update_masked): only covered voxels contribute;
        mean keeps a running sum, min / max the extreme; `count` the number of contributions."""
        from pytorch_connectomics_amd.inference.tta import _MODE_CODE
        inside = torch.ones_like(x, dtype=torch.bool) if cover is None else cover > 0
        if mode == _MODE_CODE["mean"]:
            stat += torch.where(inside, x, torch.zeros_like(x))
        elif mode == _MODE_CODE["min"]:
            stat.copy_(torch.where(inside, torch.minimum(stat, x), stat))
        else:
            stat.copy_(torch.where(inside, torch.maximum(stat, x), stat))
        count += inside.to(count.dtype)

    @staticmethod
    def ensemble_finalize_masked(stat, count, out, mode):
        from pytorch_connectomics_amd.inference.tta import _MODE_CODE
        out.copy_(stat / count if mode == _MODE_CODE["mean"] else stat)


def _net_lazy(x):
    ramp = torch.linspace(0, 1, x.shape[-1]).view(1, 1, 1, 1, -1)
    return torch.cat([2 * x - 1 + ramp, 0.5 * x + x.mean(dim=(2, 3, 4), keepdim=True)], 1)


@pytest.fixture()
def cpu_kernels(monkeypatch):
    import pytorch_connectomics_amd.inference.lazy as lazy
    import pytorch_connectomics_amd.inference.tta as tta
    import pytorch_connectomics_amd.inference.tta_ensemble as ens
    for mod in (lazy, tta, ens):
        monkeypatch.setattr(mod, "ops", _Ops)
    return lazy


@pytest.mark.parametrize("name", list(LAZY_TTA_CASES))
def test_lazy_tta_and_mask_match_the_reference_loop(name, golden_dir, cpu_kernels):
    g = np.load(golden_dir / "lazy_tta.npz")
    case = LAZY_TTA_CASES[name]
    cfg = lazy_tta_cfg(**case["cfg"])
    kw = dict(mask_path=g["mask"] if case.get("mask") else None, device="cpu")
    if case.get("region") is None:
        y = cpu_kernels.lazy_predict_volume(cfg, _net_lazy, g["vol"], **kw)
    else:
        y = cpu_kernels.lazy_predict_region(cfg, _net_lazy, image_path=g["vol"], region_start=case["region"][0],
                                            region_stop=case["region"][1], **kw)
    want = g[f"{name}__y"]
    assert tuple(y.shape) == want.shape
    np.testing.assert_allclose(y.numpy(), want, rtol=2e-5, atol=2e-5)


def test_lazy_entry_points_take_the_reference_keywords(golden_dir, cpu_kernels):
    g = np.load(golden_dir / "lazy_tta.npz")
    cfg = lazy_tta_cfg(roi=(8, 12, 16), tta=False)
    a = cpu_kernels.lazy_predict_volume(cfg, _net_lazy, g["vol"], device="cpu")
    b = cpu_kernels.lazy_predict_volume(cfg, _net_lazy, image_path=g["vol"], mask_path=None, mask_align_to_image=False, device="cpu")
    c = cpu_kernels.lazy_predict_volume(cfg, _net_lazy, volume=g["vol"], device="cpu")          # this package's earlier keyword
    assert torch.equal(a, b) and torch.equal(a, c)
    with pytest.raises(TypeError, match="pass the test volume once"):
        cpu_kernels.lazy_predict_volume(cfg, _net_lazy, device="cpu")
    with pytest.raises(ValueError, match="mask volume shape"):
        cpu_kernels.lazy_predict_volume(cfg, _net_lazy, g["vol"], mask_path=g["mask"][:, :10], device="cpu")
    # both argument orders of get_lazy_image_reference_shape
    assert cpu_kernels.get_lazy_image_reference_shape(g["vol"], cfg) == (20, 30, 34)
    assert cpu_kernels.get_lazy_image_reference_shape(torch.zeros(1, 2, 4, 5, 6)) == (4, 5, 6)


def test_reference_shape_of_a_stored_volume(tmp_path):
    """`get_lazy_image_reference_shape(cfg, image_path)` in the reference's argument order: (1, C, *padded shape); a transformed
    volume smaller than data.dataloader.patch_size is refused with the reference's message."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.inference.lazy import get_lazy_image_reference_shape
    np.save(tmp_path / "v.npy", np.zeros((2, 9, 10, 11), np.uint8))
    cfg = NS(data=NS(dataloader=NS(patch_size=[8, 8, 8]), data_transform=NS(pad_size=[1, 2, 3], pad_mode="reflect", val_transpose=None,
                                                                         resize=None),
                     image_transform=NS(normalize="none", clip_percentile_low=0.0, clip_percentile_high=1.0, resize=None),
                     mask_transform=None),
             system=NS(num_workers=1))
    assert get_lazy_image_reference_shape(cfg, str(tmp_path / "v.npy")) == (1, 2, 11, 14, 17)
    assert get_lazy_image_reference_shape(str(tmp_path / "v.npy"), cfg) == (11, 14, 17)          # earlier form of this package
    cfg.data.dataloader.patch_size = [16, 8, 8]
    with pytest.raises(ValueError, match=r"at least as large as data.dataloader.patch_size in every axis. Got transformed_shape=\(9, 10, 11\)"):
        get_lazy_image_reference_shape(cfg, str(tmp_path / "v.npy"), mode="test")


def test_chunked_runner_passes_mask_and_views_to_every_chunk(tmp_path, golden_dir, cpu_kernels):
    """run_chunked_prediction_inference under the reference's keywords (`image_path=`, `mask_path=`, `qc_streaming_callback=`):
    the stitched chunks equal the whole-volume lazy prediction with the same per-window TTA and mask."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.inference.chunked import run_chunked_prediction_inference
    g = np.load(golden_dir / "lazy_tta.npz")
    cfg = lazy_tta_cfg(roi=(8, 12, 16), flips=[[1]], acts=[{"channels": "0", "activation": "sigmoid"}], padding_mode="constant")
    cfg.inference.chunking = NS(enabled=True, chunk_size=[9, 16, 20], halo=[2, 3, 4], axes="all", shard_id=None, num_shards=None)
    full = cpu_kernels.lazy_predict_volume(cfg, _net_lazy, g["vol"], mask_path=g["mask"], device="cpu")
    seen = []
    probe = NS(update=lambda a, z_offset, z_axis: seen.append((a.shape, z_offset, z_axis)))
    out = run_chunked_prediction_inference(cfg, _net_lazy, image_path=g["vol"], mask_path=g["mask"], output_path=tmp_path / "p.npy",
                                           device="cpu", qc_streaming_callback=probe)
    np.testing.assert_allclose(out, full[0].numpy(), rtol=0, atol=0)
    assert seen == [(out.shape, 0, 1)]
    unmasked = cpu_kernels.lazy_predict_volume(cfg, _net_lazy, g["vol"], device="cpu")
    assert not torch.equal(unmasked, full)                                   # the mask did reach the windows
    with pytest.raises(TypeError, match="needs the test volume"):
        run_chunked_prediction_inference(cfg, _net_lazy, output_path=tmp_path / "q.npy", device="cpu")


def test_window_contract_messages_and_batch_guard():
    """A network that changes the spatial shape of its window is refused with the reference's messages in the caller's (N, C, Z, Y, X)
    order (reference lazy.py:389-419); a batch whose size is not the number of window positions is refused before any kernel runs."""
    from pytorch_connectomics_amd import hip_ops
    from pytorch_connectomics_amd.inference.lazy import _require_window_shape
    _require_window_shape(torch.zeros(2, 3, 6, 8, 8), (6, 8, 8), (6, 8, 8), (0, 0, 0))
    _require_window_shape(torch.zeros(2, 6, 8, 8, 3), (6, 8, 8), (6, 8, 8), (0, 0, 0), channels_last=True)
    with pytest.raises(RuntimeError, match=r"Got prediction.shape=\(2, 3, 5, 8, 8\) and roi_size=\(6, 8, 8\)\."):
        _require_window_shape(torch.zeros(2, 5, 8, 8, 3), (6, 8, 8), (6, 8, 8), (0, 0, 0), channels_last=True)
    with pytest.raises(RuntimeError, match=r"Got prediction.shape=\(2, 6, 8, 8\) and roi_size"):
        _require_window_shape(torch.zeros(2, 6, 8, 8), (6, 8, 8), (6, 8, 8), (0, 0, 0))
    with pytest.raises(RuntimeError, match=r"with target_context=\(1, 1, 1\) expected prediction spatial shape \(8, 10, 10\), got \(6, 8, 8\)\."):
        _require_window_shape(torch.zeros(2, 3, 6, 8, 8), (8, 10, 10), (6, 8, 8), (1, 1, 1))
    assert len(hip_ops._starts_array([(0, 0, 0), (1, 2, 3)], 2)) == 6
    with pytest.raises(ValueError, match="1 window predictions for 2 window positions"):
        hip_ops._starts_array([(0, 0, 0), (1, 2, 3)], 1)


def test_lazy_loop_and_window_predictor_run_the_network_without_autograd(golden_dir, cpu_kernels):
    """`@torch.no_grad()` had slid onto a helper, so the lazy loop ran the network with autograd enabled (the HIP
    models then take their training forward: saved activations, other kernels).  The reference decorates the loop itself
    (lazy.py:986) -- a caller without an outer no_grad must still meet a grad-free forward."""
    g = np.load(golden_dir / "lazy_tta.npz")
    seen = []

    def net(x):
        seen.append(torch.is_grad_enabled())
        return _net_lazy(x)

    assert torch.is_grad_enabled()
    for tta in (False, True):
        cfg = lazy_tta_cfg(roi=(8, 12, 16), tta=tta)
        cpu_kernels.lazy_predict_volume(cfg, net, g["vol"], device="cpu")
    assert seen and not any(seen)
    assert hasattr(cpu_kernels._lazy_sliding_window, "__wrapped__")
