"""GPU parity: sliding-window kernels and the on-device eager engine vs the oracle / reference fixtures."""
import numpy as np
import pytest
import torch

from oracle import window_oracle as WO

pytestmark = pytest.mark.gpu

NETS = {
    "identity": lambda x: x,
    "patch_mean": lambda x: x + x.mean(dim=(2, 3, 4), keepdim=True),
    "two_channel": lambda x: torch.cat([x * 0.5 + torch.linspace(0, 1, x.shape[-1], device=x.device).view(1, 1, 1, 1, -1),
                                        torch.tanh(x) - 0.25 * x.mean(dim=(2, 3, 4), keepdim=True)], 1),
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def test_gather_windows_matches_oracle(dev, golden_dir):
    from pytorch_connectomics_amd import hip_ops as ops
    g = np.load(golden_dir / "extract_patch.npz")
    x = torch.from_numpy(g["x"])
    vol = x[0].to(dev).contiguous()
    for i in range(int(g["n"])):
        start, roi, mode = tuple(g[f"start_{i}"]), tuple(g[f"roi_{i}"]), str(g[f"mode_{i}"])
        from pytorch_connectomics_amd.inference.window import _effective_pad_mode
        eff = _effective_pad_mode(start, roi, tuple(x.shape[2:]), mode)
        out = ops.gather_windows(vol, [start], roi, pad_mode=eff, cval=0.25)
        got = out.permute(0, 4, 1, 2, 3).cpu().numpy()
        np.testing.assert_array_equal(got, g[f"patch_{i}"])


@pytest.mark.parametrize("view", [0, 1, 2, 4, 7, 8, 13])
def test_gather_and_blend_views_roundtrip(dev, view):
    """view(gather) then blend with the same view must land every voxel where it came from."""
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd import _native as nat
    torch.manual_seed(view)
    vol = torch.rand(2, 9, 12, 12, device=dev)
    roi = (5, 8, 8)
    starts = [(1, 2, 3), (4, 4, 4)]
    win = ops.gather_windows(vol, starts, roi, view=view)
    # reference semantics on the CPU: flips then optional y/x swap
    ref = []
    for s in starts:
        w = vol[:, s[0]:s[0] + 5, s[1]:s[1] + 8, s[2]:s[2] + 8].cpu()
        if view & 8:
            w = w.transpose(2, 3)
        dims = [d + 1 for d, bit in enumerate((1, 2, 4)) if view & bit]
        if dims:
            w = torch.flip(w, dims)
        ref.append(w.permute(1, 2, 3, 0))
    assert torch.equal(win.cpu(), torch.stack(ref))
    value = torch.zeros_like(vol)
    weight = torch.zeros(vol.shape[1:], device=dev)
    ones = [torch.ones(n, device=dev) for n in roi]
    ops.blend_accumulate(win, starts, value, weight, *ones, view=view, combine=nat.BLEND_PRODUCT, floor_w=0.0)
    ops.blend_finalize(value, weight, clamp=1e-4)
    covered = weight > 0
    assert torch.equal(value[:, covered], vol[:, covered])


def test_eager_engine_matches_reference_fixtures(dev, golden_dir):
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    g = np.load(golden_dir / "eager_engine.npz")
    for name in g["names"]:
        mode, pmode, net, swb = g[f"{name}__meta"]
        eng = EagerSlidingWindowEngine(roi_size=tuple(g[f"{name}__roi"]), sw_batch_size=int(swb),
                                       overlap=tuple(g[f"{name}__ov"]), mode=str(mode), padding_mode=str(pmode),
                                       cval=0.0)
        y = eng(torch.from_numpy(g[f"{name}__x"]).to(dev), NETS[str(net)]).cpu().numpy()
        exp = g[f"{name}__y"]
        assert y.shape == exp.shape, name
        if str(net) == "identity" and str(mode) != "bump":
            # same fp32 op order as the reference -> bit identical (the bump table goes through the host's
            # libm exp, which may differ in the last ulp between the fixture host and this one)
            np.testing.assert_array_equal(y, exp, err_msg=name)
        else:
            np.testing.assert_allclose(y, exp, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(exp).max()), err_msg=name)


def test_eager_engine_determinism_and_identity_property(dev):
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    x = torch.rand(1, 1, 70, 90, 100, device=dev)
    eng = EagerSlidingWindowEngine(roi_size=(32, 32, 32), sw_batch_size=8, overlap=0.5, mode="bump",
                                   padding_mode="constant", cval=0.0)
    a = eng(x, lambda t: t)
    b = eng(x, lambda t: t)
    assert torch.equal(a, b)                     # race-free by construction
    # identity network + weighted mean reproduces the input wherever the accumulated weight clears the
    # 1e-4 normalisation floor (volume faces are deliberately driven towards 0, window.py:275-294)
    assert torch.allclose(a[..., 4:-4, 4:-4, 4:-4], x[..., 4:-4, 4:-4, 4:-4], atol=1e-5)
    ref = WO.eager_sliding_window(x.cpu(), lambda t: t, roi=(32, 32, 32), overlap=0.5, mode="bump", sw_batch_size=8)
    # the bump table goes through the host libm (numpy in the oracle, torch in the product): last-ulp slack
    torch.testing.assert_close(a.cpu(), ref, rtol=2e-6, atol=1e-7)
    eng_c = EagerSlidingWindowEngine(roi_size=(32, 32, 32), sw_batch_size=8, overlap=0.5, mode="constant",
                                     padding_mode="constant", cval=0.0)
    ref_c = WO.eager_sliding_window(x.cpu(), lambda t: t * 3 - 1, roi=(32, 32, 32), overlap=0.5, mode="constant",
                                    sw_batch_size=8)
    assert torch.equal(eng_c(x, lambda t: t * 3 - 1).cpu(), ref_c)   # same fp32 op order -> bit exact


def test_normalize_matches_reference(dev, golden_dir):
    from pytorch_connectomics_amd.inference.window import normalize_weighted_accumulator
    g = np.load(golden_dir / "normalize.npz")
    v = torch.from_numpy(g["value"]).to(dev)
    w = torch.from_numpy(g["weight"]).to(dev)
    out = normalize_weighted_accumulator(v, w)
    np.testing.assert_array_equal(out.cpu().numpy(), g["expected"])


def test_ensemble_update(dev):
    from pytorch_connectomics_amd import hip_ops as ops
    xs = [torch.rand(1000, device=dev) for _ in range(5)]
    for mode, ref in ((1, torch.stack(xs).min(0).values), (2, torch.stack(xs).max(0).values)):
        acc = torch.empty(1000, device=dev)
        for i, x in enumerate(xs):
            ops.ensemble_update(acc, x, mode, i + 1)
        assert torch.equal(acc, ref)
    acc = torch.empty(1000, device=dev)
    cpu = None
    for i, x in enumerate(xs):
        ops.ensemble_update(acc, x, 0, i + 1)
        cpu = x.cpu().clone() if cpu is None else cpu + (x.cpu() - cpu) / (i + 1)   # tta_ensemble.py:95-97
    assert torch.equal(acc.cpu(), cpu)


@pytest.mark.parametrize("mode,np_mode", [("reflect", "reflect"), ("replicate", "edge"), ("circular", "wrap"),
                                          ("constant", "constant")])
def test_gather_outer_padding_follows_numpy_pad_of_the_inner_crop(dev, mode, np_mode):
    """Windows of the lazy grid overhang the volume by up to roi/2; the reference pads the in-volume crop of
    each window with np.pad (lazy.py:852-904) -- incl. the periodic bounce when the pad reaches the crop size."""
    import itertools
    from pytorch_connectomics_amd import hip_ops as ops
    rng = np.random.default_rng(1)
    vol = rng.random((2, 9, 10, 14), dtype=np.float32)
    roi = (4, 6, 8)
    offs = WO.lazy_axis_offsets(vol.shape[1:], roi, 0.5)
    wins = list(itertools.product(*offs))
    got = ops.gather_windows(torch.from_numpy(vol).to(dev), wins, roi, pad_mode=mode, cval=0.5).cpu().numpy()
    for i, w in enumerate(wins):
        lo = [max(0, w[a]) for a in range(3)]
        hi = [min(vol.shape[1 + a], w[a] + roi[a]) for a in range(3)]
        inner = vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        pads = [(0, 0)] + [(max(0, -w[a]), max(0, w[a] + roi[a] - vol.shape[1 + a])) for a in range(3)]
        kw = dict(constant_values=0.5) if np_mode == "constant" else {}
        m = np_mode
        if m == "reflect" and min(inner.shape[1:]) <= 1:
            m = "edge"
        ref = np.pad(inner, pads, mode=m, **kw)
        np.testing.assert_array_equal(got[i], np.moveaxis(ref, 0, -1), err_msg=f"window {w}")


@pytest.mark.parametrize("target,scale", [("uint8", 255.0), ("int8", 100.0), ("uint16", 65535.0), ("int16", 3e4),
                                          ("int32", 1e6), ("float16", 1.0), ("float32", 2.5), ("uint8", -1.0)])
def test_scale_cast_matches_numpy(target, scale):
    """Device prediction / storage transform = the reference's numpy ops (scale in fp32, clip, truncating cast)."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.inference.output import apply_prediction_transform, apply_storage_dtype_transform
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(1, 2, 9, 10, 11, generator=g) * 1.6 - 0.3)
    cfg = NS(inference=NS(prediction_transform=NS(enabled=True, intensity_scale=scale, intensity_dtype=target), save_dtype=None))
    got = apply_prediction_transform(cfg, x.cuda())
    want = apply_prediction_transform(cfg, x.numpy().copy())
    assert str(got.dtype).replace("torch.", "") == str(want.dtype)
    assert np.array_equal(got.cpu().numpy(), want)
    st = apply_storage_dtype_transform(NS(inference=NS(save_dtype="float16")), x.cuda())
    assert np.array_equal(st.cpu().numpy(), x.numpy().astype(np.float16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_window_pipeline_streams_do_not_change_the_result(dev, dtype):
    """Window batches spread over 1 / 2 / 4 HIP streams (EagerSlidingWindowEngine.pipeline_streams): the accumulators see
    the batches in window order whatever the stream count (reference window.py:648-675), and no kernel of the forward may
    change its result when another batch's kernels share the GPU -- round 3 found one that did (the K = 3 x-block
    depthwise conv, compiler-generated packed-fp32 FMAs; csrc/build.py EXTRA_FLAGS)."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    from pytorch_connectomics_amd.models import build_model
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=1, mednext=NS(size="S", kernel_size=3),
                      loss=NS(deep_supervision=False), heads=None))
    torch.manual_seed(0)
    model = build_model(cfg).to(dev).eval()
    model.model.compute_dtype = dtype
    shape, swb = ((165, 280, 336), 4) if dtype == torch.bfloat16 else ((112, 224, 280), 2)
    vol = torch.rand((1, 1) + shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    eng = EagerSlidingWindowEngine(roi_size=(112, 112, 112), sw_batch_size=swb, overlap=0.5, mode="bump",
                                   padding_mode="constant", cval=0.0)
    assert (len(eng.plan(shape)[1]) - 1 + swb - 1) // swb >= 4          # enough window batches to fill four streams
    outs = {}
    for n in (1, 2, 4, 2):
        eng.pipeline_streams = n
        with torch.no_grad():
            y = eng(vol, model)
        torch.cuda.synchronize()
        assert eng.last_stats["streams"] == n
        outs.setdefault(n, []).append(y)
    ref = outs[1][0]
    for n, ys in outs.items():
        for y in ys:
            assert torch.equal(ref, y), f"{n} streams changed the result (max |d| {float((ref - y).abs().max())})"


@pytest.mark.parametrize("C,c0,c1,act", [(7, 0, 7, "sigmoid"), (7, 3, 6, "sigmoid"), (4, 0, 4, "tanh"), (3, 1, 2, "sigmoid"), (8, 0, 8, "none")])
def test_channels_last_activation_flat_form_equals_the_per_voxel_form(C, c0, c1, act):
    """pytc_channel_activation on channels-last predictions (the lazy loop's per-batch activation): the flat float4 kernel against the
    per-voxel kernel (knob channel_act_flat) -- same expressions per element, identical bits; untouched channels stay untouched."""
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd import _native as nat
    code = {"sigmoid": nat.ACT_SIGMOID, "tanh": nat.ACT_TANH, "none": nat.ACT_NONE}[act]
    x = (torch.randn(2, 12, 10, 14, C, generator=torch.Generator().manual_seed(C + c0)) * 4).cuda()
    a, b = x.clone(), x.clone()
    ops.channel_activation(a, c0, c1, code, 1.5, channels_last=True)
    ops.set_tuning("channel_act_flat", 0)
    try:
        ops.channel_activation(b, c0, c1, code, 1.5, channels_last=True)
    finally:
        ops.set_tuning("channel_act_flat", 1)
    assert torch.equal(a, b)
    keep = [c for c in range(C) if not (c0 <= c < c1)]
    assert torch.equal(a[..., keep], x[..., keep])
    ref = x[..., c0:c1] * 1.5
    ref = torch.sigmoid(ref) if act == "sigmoid" else (torch.tanh(ref) if act == "tanh" else ref)
    assert torch.allclose(a[..., c0:c1], ref, atol=1e-6, rtol=1e-5)
