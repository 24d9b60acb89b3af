"""Generate golden fixtures from the REFERENCE's own modules (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference through tests/golden/_ref_shim.py, feeds seeded inputs through the
reference functions of the hot path and stores inputs + expected outputs as small .npz files
next to this script.  The fixtures are data; no reference source travels with them.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _ref_shim as S  # noqa: E402

win = S.ref("connectomics.inference.window")
rsunet = S.ref("connectomics.models.architectures.rsunet")
cgrid = S.ref("connectomics.chunked.chunk_grid")
chalo = S.ref("connectomics.chunked.halo")


def save(name, **arrs):
    np.savez_compressed(HERE / name, **arrs)
    print("wrote", name, len(arrs), "arrays")


# ---------------------------------------------------------------- window grids
def grids():
    cases = [
        ((165, 1024, 768), (112, 112, 112), 0.5),     # Lucchi++ test volume (C2)
        ((448, 448, 448), (112, 112, 112), 0.5),      # cubic bench volume
        ((640, 640, 640), (160, 160, 160), 0.5),      # C4
        ((100, 1024, 1024), (32, 160, 160), 0.5),     # SNEMI-like anisotropic
        ((128, 128, 128), (64, 64, 64), 0.5),         # C1 inference pass
        ((4, 5, 6), (2, 3, 3), 0.5),                  # reference lazy test volume
        ((24, 24, 24), (16, 16, 16), 0.5),
        ((24, 24, 24), (16, 16, 16), 0.0),
        ((32, 32, 32), (64, 64, 64), 0.0),            # image smaller than roi
        ((30, 70, 50), (16, 32, 24), (0.25, 0.5, 0.75)),
        ((17, 33, 9), (8, 8, 16), 0.99),              # overlap clamp, stride 1 + img<roi axis
        ((50, 50, 50), (7, 11, 13), 0.3),
    ]
    out = {}
    for i, (img, roi, ov) in enumerate(cases):
        iv = win.compute_scan_interval(img, roi, overlap=ov)
        st = win.dense_patch_slices(img, roi, iv, return_slice=False)
        out[f"img_{i}"] = np.asarray(img, np.int64)
        out[f"roi_{i}"] = np.asarray(roi, np.int64)
        out[f"ov_{i}"] = np.asarray(ov if isinstance(ov, tuple) else (ov,) * 3, np.float64)
        out[f"interval_{i}"] = np.asarray(iv, np.int64)
        out[f"starts_{i}"] = np.asarray(st, np.int64)
    out["n"] = np.asarray(len(cases))
    save("window_grids.npz", **out)


# ---------------------------------------------------------------- importance maps
def maps():
    out = {}
    for mode in ("constant", "bump", "distance_transform"):
        for roi in ((8, 8, 8), (5, 5, 5), (2, 3, 3), (4, 6, 10)):
            m = win.build_sliding_importance_map(roi, mode=mode, device="cpu", dtype=torch.float32)
            out[f"{mode}_{'x'.join(map(str, roi))}"] = m.numpy()
    big = win.build_sliding_importance_map((112, 112, 112), mode="bump", device="cpu", dtype=torch.float32)
    out["bump_112_z"] = big[:, 56, 56].numpy()
    out["bump_112_y"] = big[56, :, 56].numpy()
    out["bump_112_x"] = big[56, 56, :].numpy()
    out["bump_112_corner"] = big[:4, :4, :4].numpy()
    out["bump_112_diag"] = torch.stack([big[i, i, i] for i in range(112)]).numpy()
    raw = win.compute_importance_map((8, 8, 8), mode="bump")
    out["bump_raw_8x8x8"] = raw.numpy()
    save("importance_maps.npz", **out)


# ---------------------------------------------------------------- normalisation
def normalise():
    g = torch.Generator().manual_seed(3)
    val = torch.rand(1, 2, 6, 7, 8, generator=g)
    wgt = torch.rand(1, 1, 6, 7, 8, generator=g)
    wgt[0, 0, 0] = 1e-7          # below the 1e-4 clamp
    wgt[0, 0, 1, 0] = 0.0
    wgt[0, 0, 2] = 5e-5
    exp = win.normalize_weighted_accumulator(val.clone(), wgt.clone())
    v16, w16 = val.half(), wgt.half()
    exp16 = win.normalize_weighted_accumulator(v16.clone(), w16.clone())
    save("normalize.npz", value=val.numpy(), weight=wgt.numpy(), expected=exp.numpy(),
         value16=v16.numpy(), weight16=w16.numpy(), expected16=exp16.numpy())


# ---------------------------------------------------------------- eager engine
def _net_identity(x):
    return x


def _net_patch_mean(x):        # reference tests/unit/test_lazy_inference.py:28-29
    return x + x.mean(dim=(2, 3, 4), keepdim=True)


def _net_two_channel(x):       # position dependent, 2 output channels
    ramp = torch.linspace(0, 1, x.shape[-1]).view(1, 1, 1, 1, -1)
    return torch.cat([x * 0.5 + ramp, torch.tanh(x) - 0.25 * x.mean(dim=(2, 3, 4), keepdim=True)], 1)


NETS = {"identity": _net_identity, "patch_mean": _net_patch_mean, "two_channel": _net_two_channel}


def engine():
    cases = [
        # name, image shape, roi, overlap, mode, padding_mode, net, sw_batch
        ("arange24_const", (24, 24, 24), (16, 16, 16), 0.5, "constant", "constant", "identity", 4),
        ("arange24_noov", (24, 24, 24), (16, 16, 16), 0.0, "constant", "constant", "identity", 4),
        ("small_lt_roi", (12, 12, 12), (16, 16, 16), 0.0, "constant", "constant", "identity", 1),
        ("small_lt_roi_reflect", (12, 12, 12), (16, 16, 16), 0.0, "constant", "reflect", "identity", 1),
        ("bump_pm", (20, 27, 31), (8, 12, 16), 0.5, "bump", "constant", "patch_mean", 3),
        ("bump_2ch", (20, 27, 31), (8, 12, 16), 0.5, "bump", "constant", "two_channel", 8),
        ("dist_2ch", (19, 21, 40), (8, 8, 16), (0.25, 0.5, 0.5), "distance_transform", "constant", "two_channel", 2),
        ("lazy_vol", (4, 5, 6), (2, 3, 3), 0.5, "bump", "constant", "identity", 2),
        ("mixed_axes", (6, 40, 40), (8, 16, 16), 0.5, "bump", "reflect", "patch_mean", 5),
    ]
    out = {}
    names = []
    for name, shp, roi, ov, mode, pmode, net, swb in cases:
        g = torch.Generator().manual_seed(sum(ord(c) for c in name))
        if name.startswith("arange") or name == "lazy_vol":
            x = torch.arange(int(np.prod(shp)), dtype=torch.float32).reshape(1, 1, *shp)
        else:
            x = torch.rand(1, 1, *shp, generator=g)
        eng = win.EagerSlidingWindowEngine(roi_size=roi, sw_batch_size=swb, overlap=ov, mode=mode,
                                           padding_mode=pmode, cval=0.0, sw_device=None,
                                           output_device=None)
        y = eng(x, NETS[net])
        out[f"{name}__x"] = x.numpy()
        out[f"{name}__y"] = y.numpy()
        out[f"{name}__roi"] = np.asarray(roi)
        out[f"{name}__ov"] = np.asarray(ov if isinstance(ov, tuple) else (ov,) * 3, np.float64)
        out[f"{name}__meta"] = np.asarray([mode, pmode, net, str(swb)])
        names.append(name)
    out["names"] = np.asarray(names)
    save("eager_engine.npz", **out)

    # patch extraction with padding
    x = torch.rand(1, 2, 9, 10, 11, generator=torch.Generator().manual_seed(5))
    ex = {}
    for i, (start, roi, pm) in enumerate([((-2, 0, 3), (6, 6, 6), "constant"), ((5, 6, 7), (6, 6, 6), "reflect"),
                                          ((-1, -1, -1), (4, 12, 13), "replicate"), ((0, 0, 0), (20, 4, 4), "reflect")]):
        sl = tuple(slice(s, s + r) for s, r in zip(start, roi))
        p, loc = win._extract_padded_patch_batch(x, [sl], roi_size=roi, padding_mode=pm, cval=0.25)
        ex[f"start_{i}"] = np.asarray(start)
        ex[f"roi_{i}"] = np.asarray(roi)
        ex[f"mode_{i}"] = np.asarray(pm)
        ex[f"patch_{i}"] = p.numpy()
    ex["x"] = x.numpy()
    ex["n"] = np.asarray(4)
    save("extract_patch.npz", **ex)


# ---------------------------------------------------------------- RSUNet
def rsunets():
    cfgs = {
        "c1_group": dict(width=[8, 16], down_factors=[(2, 2, 2)], norm="group", num_groups=8, activation="relu"),
        "aniso_inst_elu_ds": dict(width=[6, 8, 12], norm="instance", activation="elu", deep_supervision=True),
        "batch_prelu_2d": dict(width=[4, 8, 8], norm="batch", activation="prelu", depth_2d=1, init=0.1),
    }
    for name, kw in cfgs.items():
        torch.manual_seed(11)
        m = rsunet.RSUNet(1, 2, **kw)
        if kw["norm"] == "batch":   # make running stats non-trivial
            m.train()
            with torch.no_grad():
                m(torch.randn(2, 1, 8, 16, 16))
        m.eval()
        x = torch.randn(1, 1, 12, 24, 24, generator=torch.Generator().manual_seed(12))
        with torch.no_grad():
            y = m(x)
        arrs = {"x": x.numpy()}
        for k, v in m.state_dict().items():
            arrs["sd__" + k] = v.numpy()
        if isinstance(y, dict):
            for k, v in y.items():
                arrs["y__" + k] = v.numpy()
        else:
            arrs["y__output"] = y.numpy()
        arrs["n_params"] = np.asarray(sum(p.numel() for p in m.parameters()))
        save(f"rsunet_{name}.npz", **arrs)

# ---------------------------------------------------------------- RSUNet training step (reference module + torch autograd)
def rsunet_train():
    """One forward + backward of the REFERENCE RSUNet in train() mode (BatchNorm with batch statistics): loss, every
    parameter gradient, the BatchNorm running buffers after the step.  Inputs, targets and the initial state_dict are
    stored so the HIP training path starts from the same point."""
    import torch.nn.functional as F
    cfgs = {
        "c1_group": dict(width=[8, 16], down_factors=[(2, 2, 2)], norm="group", num_groups=8, activation="relu"),
        "aniso_inst_elu_ds": dict(width=[6, 8, 12], norm="instance", activation="elu", deep_supervision=True),
        "batch_prelu_2d": dict(width=[4, 8, 8], norm="batch", activation="prelu", depth_2d=1, init=0.1),
        "none_leaky": dict(width=[4, 8], norm="none", activation="leakyrelu", negative_slope=0.05),
        # the reference's stock profile (config/profiles/arch_profiles.yaml:34-44): widths that are no multiple of 8 -> GroupNorm(3, 18),
        # GroupNorm(4, 36), ...; and the same widths with BatchNorm (tutorials/syn_cremi.yaml overrides `norm: batch`)
        # (two levels keep the fixture small; the full five-level profile is checked against the oracle at test time)
        "stock_group_elu": dict(width=[18, 36], norm="group", num_groups=4, activation="elu",
                                down_factors=[(1, 2, 2)], depth_2d=1, kernel_2d=(1, 3, 3)),
        "stock_batch_elu": dict(width=[18, 36], norm="batch", num_groups=8, activation="elu",
                                down_factors=[(1, 2, 2)], depth_2d=1, kernel_2d=(1, 3, 3)),
    }
    only = os.environ.get("PYTC_GOLDEN_RSUNET_TRAIN")           # comma-separated config names: regenerate these only
    for name, kw in cfgs.items():
        if only and name not in only.split(","):
            continue
        torch.manual_seed(21)
        m = rsunet.RSUNet(1, 2, **kw).train()
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, (torch.nn.GroupNorm, torch.nn.BatchNorm3d)):
                    mod.weight.uniform_(0.7, 1.3)
                    mod.bias.normal_(0, 0.2)
        arrs = {}
        for k, v in m.state_dict().items():
            arrs["sd__" + k] = v.detach().clone().numpy()
        x = torch.randn(2, 1, 8, 16, 16, generator=torch.Generator().manual_seed(22))
        t = torch.randn(2, 2, 8, 16, 16, generator=torch.Generator().manual_seed(23))
        x_in = x.clone()       # with norm="none" the reference's in-place activation rewrites the caller's tensor
        out = m(x)
        if isinstance(out, dict):
            loss = F.mse_loss(out["output"], t)
            for k in sorted(out):
                if k != "output":
                    loss = loss + 0.5 * out[k].pow(2).mean()
        else:
            loss = F.mse_loss(out, t)
        loss.backward()
        arrs["x"], arrs["t"], arrs["loss"] = x_in.numpy(), t.numpy(), np.asarray([float(loss)], np.float64)
        for k, p in m.named_parameters():
            arrs["grad__" + k] = p.grad.numpy()
        for k, b in m.named_buffers():
            arrs["buf__" + k] = b.detach().numpy()
        print(name, float(loss), len([k for k in arrs if k.startswith("grad__")]), "grads")
        save(f"rsunet_train_{name}.npz", **arrs)


# ---------------------------------------------------------------- chunk grid / halo
def chunks():
    out = {}
    cases = [((640, 640, 640), (320, 320, 320), (80, 80, 80), (0, 0, 0)),
             ((100, 333, 250), (64, 128, 128), (8, 16, 16), (2, 3, 4)),
             ((4, 5, 6), (4, 5, 6), (0, 0, 0), (0, 0, 0)),
             ((9, 9, 9), (4, 4, 4), (1, 2, 3), (0, 0, 0))]
    for i, (vol, ch, halo, crop) in enumerate(cases):
        refs = cgrid.build_chunk_grid(vol, ch)
        rows, keys = [], []
        in_shape = tuple(v + 2 * c for v, c in zip(vol, crop))
        for r in refs:
            rs, re, sl = chalo.resolve_halo_region(r, in_shape, halo=halo, crop_before=crop)
            rows.append(list(r.index) + list(r.start) + list(r.stop) + list(rs) + list(re)
                        + [s.start for s in sl] + [s.stop for s in sl])
            keys.append(r.key)
        out[f"vol_{i}"] = np.asarray(vol)
        out[f"chunk_{i}"] = np.asarray(ch)
        out[f"halo_{i}"] = np.asarray(halo)
        out[f"crop_{i}"] = np.asarray(crop)
        out[f"rows_{i}"] = np.asarray(rows, np.int64)
        out[f"keys_{i}"] = np.asarray(keys)
    out["n"] = np.asarray(len(cases))
    save("chunk_grid.npz", **out)




# ---------------------------------------------------------------- TTA (flip / rot90 ensembles)
def _net_asym(x):
    """Closed-form, NOT flip/rotation equivariant, 3 output channels."""
    z = torch.linspace(-1, 1, x.shape[2]).view(1, 1, -1, 1, 1)
    y = torch.linspace(-1, 1, x.shape[3]).view(1, 1, 1, -1, 1)
    w = torch.linspace(-1, 1, x.shape[4]).view(1, 1, 1, 1, -1)
    a = x * (1.0 + 0.5 * w) + 0.25 * y
    b = torch.tanh(2 * x - 1) * z + 0.1 * w * y
    c = 3 * x * x - 1.5 * w + z * y
    return torch.cat([a, b, c], 1)


def tta():
    from types import SimpleNamespace as NS
    mgr = S.ref("connectomics.inference.manager")

    def cfg_for(tta_ns, *, acts, select=None, roi=(8, 12, 12), blending="bump"):
        return NS(
            model=NS(primary_head=None, heads=None, out_channels=3),
            data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1)),
            inference=NS(
                sliding_window=NS(window_size=list(roi), sw_batch_size=3, overlap=0.5, blending=blending,
                                  padding_mode="constant", cval=0.0, keep_input_on_cpu=False, sw_device=None,
                                  output_device=None, border_mask=None, distributed_sharding=False),
                model=NS(head=None, select_channel=select, output_dtype=None, channel_activations=acts,
                         crop_pad=None),
                test_time_augmentation=tta_ns,
            ),
        )

    x = torch.rand(1, 1, 14, 22, 26, generator=torch.Generator().manual_seed(21))
    out = {"x": x.numpy()}
    cases = {
        "flip8_mean_sigmoid": (NS(enabled=True, flip_axes="all", rotation90_axes=None, rotate90_k=None,
                                  ensemble_mode="mean", patch_first_local=True, distributed_sharding=False,
                                  apply_mask=True, empty_cache_interval=0),
                               [{"channels": ":", "activation": "sigmoid"}], None),
        "rot16_min_mixed": (NS(enabled=True, flip_axes="all", rotation90_axes=[[1, 2]], rotate90_k=None,
                               ensemble_mode=[["0:2", "min"], ["2", "max"]], patch_first_local=True,
                               distributed_sharding=False, apply_mask=True, empty_cache_interval=0),
                            [{"channels": "0:2", "activation": "scale_sigmoid:0.5"}, {"channels": "2", "activation": "tanh"}],
                            None),
        "flipz_select": (NS(enabled=True, flip_axes=[[0], [1, 2]], rotation90_axes=None, rotate90_k=None,
                            ensemble_mode="mean", patch_first_local=True, distributed_sharding=False,
                            apply_mask=True, empty_cache_interval=0),
                         [{"channels": ":", "activation": "softmax"}], [2, 0]),
        "notta_tanh": (NS(enabled=False), [{"channels": "1", "activation": "tanh"}], "0:2"),
    }
    xsq = torch.rand(1, 1, 14, 24, 24, generator=torch.Generator().manual_seed(22))
    out["x_square"] = xsq.numpy()
    for name, (tta_ns, acts, select) in cases.items():
        cfg = cfg_for(tta_ns, acts=acts, select=select)
        m = mgr.InferenceManager(cfg=cfg, model=torch.nn.Identity(), forward_fn=_net_asym)
        xin = xsq if name.startswith("rot") else x
        y = m.predict_with_tta(xin.clone())
        out[f"{name}__y"] = y.numpy()
        print(name, tuple(y.shape), float(y.mean()))
        if name == "flip8_mean_sigmoid":   # the whole-volume TTA path must agree on this symmetric grid
            cfg2 = cfg_for(NS(**{**vars(tta_ns), "patch_first_local": False}), acts=acts, select=select)
            y2 = mgr.InferenceManager(cfg=cfg2, model=torch.nn.Identity(), forward_fn=_net_asym).predict_with_tta(x.clone())
            print("  whole-volume vs patch-first max diff", float((y - y2).abs().max()))
    # patch_first_local: false -- every view is its own whole-volume sliding pass (tta.py:691-769, 806-878); the window grid of
    # a flipped / rotated 14x22x26 volume differs from the identity view's, and odd rotations of a non-square plane are allowed
    whole = {
        "whole_flip8_mean_sigmoid": (NS(**{**vars(cases["flip8_mean_sigmoid"][0]), "patch_first_local": False}),
                                     [{"channels": ":", "activation": "sigmoid"}], None),
        "whole_rot16_nonsquare_minmax": (NS(**{**vars(cases["rot16_min_mixed"][0]), "patch_first_local": False}),
                                         cases["rot16_min_mixed"][1], None),
    }
    for name, (tta_ns, acts, select) in whole.items():
        cfg = cfg_for(tta_ns, acts=acts, select=select, roi=(8, 12, 16))
        y = mgr.InferenceManager(cfg=cfg, model=torch.nn.Identity(), forward_fn=_net_asym).predict_with_tta(x.clone())
        out[f"{name}__y"] = y.numpy()
        print(name, tuple(y.shape), float(y.mean()))
    save("tta.npz", **out)


# ---------------------------------------------------------------- affinity-aware TTA
def _net_aff(x, n_out):
    """Closed-form, not equivariant, n_out channels with distinct spatial structure."""
    z = torch.linspace(-1, 1, x.shape[2]).view(1, 1, -1, 1, 1)
    y = torch.linspace(-1, 1, x.shape[3]).view(1, 1, 1, -1, 1)
    w = torch.linspace(-1, 1, x.shape[4]).view(1, 1, 1, 1, -1)
    chans = [x * (1.0 + 0.5 * w) + 0.25 * y, torch.tanh(2 * x - 1) * z + 0.1 * w * y, 3 * x * x - 1.5 * w + z * y,
             x * z - 0.3 * y * w, 0.5 * x + w * w - z, torch.sin(3 * x) + 0.2 * y - 0.4 * z * w]
    return torch.cat(chans[:n_out], 1)


def tta_affinity():
    from types import SimpleNamespace as NS
    mgr = S.ref("connectomics.inference.manager")

    def cfg_for(tta_ns, *, n_out, offsets, mode, acts, select=None, roi=(8, 12, 12)):
        return NS(
            model=NS(primary_head=None, heads=None, out_channels=n_out),
            data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1),
                    label_transform=NS(stack_outputs=True, targets=[{"name": "affinity", "kwargs": {"offsets": offsets, "affinity_mode": mode}}])),
            inference=NS(
                sliding_window=NS(window_size=list(roi), sw_batch_size=3, overlap=0.5, blending="bump",
                                  padding_mode="constant", cval=0.0, keep_input_on_cpu=False, sw_device=None,
                                  output_device=None, border_mask=None, distributed_sharding=False),
                model=NS(head=None, select_channel=select, output_dtype=None, channel_activations=acts, crop_pad=None),
                test_time_augmentation=tta_ns,
            ),
        )

    def tta_ns(flip, rot, mode):
        return NS(enabled=True, flip_axes=flip, rotation90_axes=rot, rotate90_k=None, ensemble_mode=mode,
                  patch_first_local=True, distributed_sharding=False, apply_mask=True, empty_cache_interval=0)

    x = torch.rand(1, 1, 14, 22, 26, generator=torch.Generator().manual_seed(31))
    xsq = torch.rand(1, 1, 14, 24, 24, generator=torch.Generator().manual_seed(32))
    out = {"x": x.numpy(), "x_square": xsq.numpy()}
    lr = ["1-0-0", "0-1-0", "0-0-1", "3-0-0", "0-3-0", "0-0-3"]
    cases = {
        "aff6_flip8_mean_deepem": (tta_ns("all", None, "mean"), 6, lr, "deepem", [{"channels": ":", "activation": "sigmoid"}], None, x),
        "aff3_rot16_min_banis": (tta_ns("all", [[1, 2]], "min"), 3, ["1-0-0", "0-1-0", "0-0-1"], "banis",
                                 [{"channels": ":", "activation": "sigmoid"}], None, xsq),
        "aff6_flipzy_select_max": (tta_ns([[0], [1], [0, 1]], None, "max"), 6, lr, "deepem",
                                   [{"channels": ":", "activation": "sigmoid"}], [3, 0, 4], x),
    }
    for name, (ns, n_out, offsets, mode, acts, select, xin) in cases.items():
        cfg = cfg_for(ns, n_out=n_out, offsets=offsets, mode=mode, acts=acts, select=select)
        m = mgr.InferenceManager(cfg=cfg, model=torch.nn.Identity(), forward_fn=lambda t, n=n_out: _net_aff(t, n))
        y = m.predict_with_tta(xin.clone())
        out[f"{name}__y"] = y.numpy()
        print(name, tuple(y.shape), float(y.mean()))
    save("tta_affinity.npz", **out)
    # the same predictor with `patch_first_local: false`: every view is a whole-volume sliding pass, inverted (affinity channels
    # re-anchored) and ensembled with per-voxel validity (tta.py:691-769) -> tests/golden/tta_affinity_whole.npz
    whole = {"x": x.numpy(), "x_square": xsq.numpy()}
    for name, (ns, n_out, offsets, mode, acts, select, xin) in {
            "whole_aff6_flip8_mean_deepem": (tta_ns("all", None, "mean"), 6, lr, "deepem", [{"channels": ":", "activation": "sigmoid"}], None, x),
            "whole_aff3_rot_min_banis": (tta_ns([[0]], [[1, 2]], "min"), 3, ["1-0-0", "0-1-0", "0-0-1"], "banis",
                                         [{"channels": ":", "activation": "sigmoid"}], None, xsq),
            "whole_aff6_select_max": (tta_ns([[1], [2], [1, 2]], None, "max"), 6, lr, "deepem",
                                      [{"channels": ":", "activation": "sigmoid"}], [3, 0, 4], x)}.items():
        ns.patch_first_local = False
        cfg = cfg_for(ns, n_out=n_out, offsets=offsets, mode=mode, acts=acts, select=select)
        m = mgr.InferenceManager(cfg=cfg, model=torch.nn.Identity(), forward_fn=lambda t, n=n_out: _net_aff(t, n))
        y = m.predict_with_tta(xin.clone())
        whole[f"{name}__y"] = y.numpy()
        print(name, tuple(y.shape), float(y.mean()))
    save("tta_affinity_whole.npz", **whole)
    # the reference's channel-move plans for a few view sets (host integer logic; compared exactly)
    import json
    ta = S.ref("connectomics.inference.tta_affinity")
    comb = S.ref("connectomics.inference.tta_combinations")
    plans = {}
    for pname, (flip, rot, n_out, offsets, mode) in {
            "lr6_deepem_all": ("all", None, 6, lr, "deepem"),
            "unit3_banis_rot": ("all", [[1, 2]], 3, ["1-0-0", "0-1-0", "0-0-1"], "banis"),
            "diag_deepem_flips": ([[0], [2], [0, 2]], None, 4, ["1-0-0", "0-1-0", "0-0-1", "0-9-0"], "deepem")}.items():
        cfg = cfg_for(tta_ns(flip, rot, "mean"), n_out=n_out, offsets=offsets, mode=mode, acts=None)
        combos = comb.resolve_tta_augmentation_combinations(cfg.inference.test_time_augmentation, spatial_dims=3)
        plan = ta.build_affinity_tta_plan(cfg, augmentation_combinations=combos, num_raw=n_out, requested_head=None)
        plans[pname] = {"flip": flip, "rot": rot, "n_out": n_out, "offsets": offsets, "mode": mode,
                        "combos": [[list(f), None if pl is None else list(pl), int(k)] for f, pl, k in combos],
                        "views": [[[m.src, m.dst, None if m.shift is None else list(m.shift)] for m in v.moves] for v in plan.views],
                        "partial": sorted(plan.partial_channels), "shifts": sorted(list(s) for s in plan.shifts)}
    (HERE / "tta_affinity_plans.json").write_text(json.dumps(plans, indent=0))
    print("wrote tta_affinity_plans.json", {k: len(v["views"]) for k, v in plans.items()})


def tta_zplanes():
    """Round 6: quarter turns in the planes that contain z (tta_combinations.py:90-119 accepts any plane whose axes have equal image and
    window size): the reference's InferenceManager.predict_with_tta on small volumes -> tests/golden/tta_zplanes.npz."""
    from types import SimpleNamespace as NS
    mgr = S.ref("connectomics.inference.manager")

    def cfg_for(tta_ns, *, n_out, acts, roi, select=None, offsets=None, amode=None):
        cfg = NS(model=NS(primary_head=None, heads=None, out_channels=n_out),
                 data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1)),
                 inference=NS(sliding_window=NS(window_size=list(roi), sw_batch_size=3, overlap=0.5, blending="bump",
                                                padding_mode="constant", cval=0.0, keep_input_on_cpu=False, sw_device=None,
                                                output_device=None, border_mask=None, distributed_sharding=False),
                              model=NS(head=None, select_channel=select, output_dtype=None, channel_activations=acts, crop_pad=None),
                              test_time_augmentation=tta_ns))
        if offsets is not None:
            cfg.data.label_transform = NS(stack_outputs=True, targets=[{"name": "affinity", "kwargs": {"offsets": offsets, "affinity_mode": amode}}])
        return cfg

    def ns(flip, rot, mode, ks=None):
        return NS(enabled=True, flip_axes=flip, rotation90_axes=rot, rotate90_k=ks, ensemble_mode=mode, patch_first_local=True,
                  distributed_sharding=False, apply_mask=True, empty_cache_interval=0)

    g = torch.Generator().manual_seed(61)
    xc = torch.rand(1, 1, 16, 16, 16, generator=g)          # cube: every plane
    xzy = torch.rand(1, 1, 20, 20, 14, generator=g)         # z == y
    xzx = torch.rand(1, 1, 16, 22, 16, generator=g)         # z == x
    out = {"x_cube": xc.numpy(), "x_zy": xzy.numpy(), "x_zx": xzx.numpy()}
    sig = [{"channels": ":", "activation": "sigmoid"}]
    cases = {
        "cube_all32_mean": (ns("all", "all", "mean"), 3, sig, (8, 8, 8), None, None, None, xc, _net_asym),
        "zy_flipx_minmax": (ns([[2]], [[0, 1]], [["0:2", "min"], ["2", "max"]]), 3,
                            [{"channels": "0:2", "activation": "scale_sigmoid:0.5"}, {"channels": "2", "activation": "tanh"}],
                            (8, 8, 12), None, None, None, xzy, _net_asym),
        "zx_k13_select": (ns(None, [[0, 2]], "mean", ks=[1, 3]), 3, [{"channels": ":", "activation": "softmax"}], (8, 12, 8), [2, 0],
                          None, None, xzx, _net_asym),
        "aff3_zx_mean_deepem": (ns([[1]], [[0, 2]], "mean"), 3, sig, (8, 12, 8), None, ["1-0-0", "0-1-0", "0-0-1"], "deepem", xzx,
                                lambda t: _net_aff(t, 3)),
        "aff6_cube_zy_min_banis": (ns("all", [[0, 1]], "min"), 6, sig, (8, 8, 8), None,
                                   ["1-0-0", "0-1-0", "0-0-1", "3-0-0", "0-3-0", "0-0-3"], "banis", xc, lambda t: _net_aff(t, 6)),
    }
    for name, (tta_ns, n_out, acts, roi, select, offsets, amode, xin, net) in cases.items():
        cfg = cfg_for(tta_ns, n_out=n_out, acts=acts, roi=roi, select=select, offsets=offsets, amode=amode)
        y = mgr.InferenceManager(cfg=cfg, model=torch.nn.Identity(), forward_fn=net).predict_with_tta(xin.clone())
        out[f"{name}__y"] = y.numpy()
        print(name, tuple(y.shape), float(y.mean()))
    save("tta_zplanes.npz", **out)
    # the reference's channel-move plans for view sets with z planes, appended to tests/golden/tta_affinity_plans.json (the host
    # test walks every entry of that file)
    import json
    ta = S.ref("connectomics.inference.tta_affinity")
    comb = S.ref("connectomics.inference.tta_combinations")
    path = HERE / "tta_affinity_plans.json"
    plans = json.loads(path.read_text())
    lr = ["1-0-0", "0-1-0", "0-0-1", "3-0-0", "0-3-0", "0-0-3"]
    for pname, (flip, rot, n_out, offsets, mode) in {
            "z_unit3_deepem_all_planes": ("all", "all", 3, ["1-0-0", "0-1-0", "0-0-1"], "deepem"),
            "z_lr6_banis_zy": ("all", [[0, 1]], 6, lr, "banis"),
            "z_lr6_deepem_zx_flipy": ([[1]], [[0, 2]], 6, lr, "deepem")}.items():
        cfg = cfg_for(ns(flip, rot, "mean"), n_out=n_out, acts=None, roi=(8, 8, 8), offsets=offsets, amode=mode)
        combos = comb.resolve_tta_augmentation_combinations(cfg.inference.test_time_augmentation, spatial_dims=3)
        plan = ta.build_affinity_tta_plan(cfg, augmentation_combinations=combos, num_raw=n_out, requested_head=None)
        plans[pname] = {"flip": flip, "rot": rot, "n_out": n_out, "offsets": offsets, "mode": mode,
                        "combos": [[list(f), None if pl is None else list(pl), int(k)] for f, pl, k in combos],
                        "views": [[[m.src, m.dst, None if m.shift is None else list(m.shift)] for m in v.moves] for v in plan.views],
                        "partial": sorted(plan.partial_channels), "shifts": sorted(list(s) for s in plan.shifts)}
    path.write_text(json.dumps(plans, indent=0))
    print("tta_affinity_plans.json now holds", sorted(plans))


def public_adapters():
    """Fixtures for the public names a drop-in importer reaches for: the reference's own `invert_view`,
    `TTAEnsembleAccumulator` and `resolve_output_head(s)` on small seeded inputs."""
    import json
    from types import SimpleNamespace as NS
    ta = S.ref("connectomics.inference.tta_affinity")
    te = S.ref("connectomics.inference.tta_ensemble")
    comb = S.ref("connectomics.inference.tta_combinations")
    mo = S.ref("connectomics.utils.model_outputs")
    lr = ["1-0-0", "0-1-0", "0-0-1", "3-0-0", "0-2-0", "0-0-2"]
    cfg = NS(model=NS(primary_head=None, heads=None, out_channels=6),
             data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1),
                     label_transform=NS(stack_outputs=True, targets=[{"name": "affinity", "kwargs": {"offsets": lr, "affinity_mode": "deepem"}}])),
             inference=NS(model=NS(head=None, select_channel=None, output_dtype=None, channel_activations=None, crop_pad=None),
                          test_time_augmentation=NS(enabled=True, flip_axes="all", rotation90_axes=[[1, 2]], rotate90_k=None,
                                                    ensemble_mode="mean")))
    combos = comb.resolve_tta_augmentation_combinations(cfg.inference.test_time_augmentation, spatial_dims=3)
    plan = ta.build_affinity_tta_plan(cfg, augmentation_combinations=combos, num_raw=6, requested_head=None)
    g = torch.Generator().manual_seed(41)
    out = {"combos": np.asarray([[sum(1 << a for a in f), -1 if pl is None else pl[0] * 3 + pl[1], k] for f, pl, k in combos], np.int64)}
    modes = ["mean", "min", "max", "mean", "max", "min"]
    acc = te.TTAEnsembleAccumulator((1, 6, 5, 8, 8), dtype=torch.float32, device=torch.device("cpu"), mode_map=modes,
                                    partial_channels=sorted(plan.partial_channels), distributed_sharding=False, max_views=len(combos))
    for i, (f, pl, k) in enumerate(combos):
        pred = torch.rand(1, 6, 5, 8, 8, generator=g)
        # the view as the network would have seen it: forward transform of a canonical tensor
        view = comb_apply(pred, f, pl, k)
        inv, val = ta.invert_view(view, flip_axes=f, rotation_plane_spatial=pl, k=k, view_plan=plan.views[i], tta_plan=plan)
        out[f"view{i}"] = view.numpy()
        out[f"inv{i}"] = inv.numpy()
        out[f"valid{i}"] = np.asarray([[-1] * 6 if v is None else [s.start for s in v] + [s.stop for s in v] for v in val.channels], np.int64)
        acc.add(inv, val)
    out["ensemble"] = acc.finalize().numpy()
    out["partial"] = np.asarray(sorted(plan.partial_channels), np.int64)
    save("public_adapters.npz", **out)
    heads = {"aff": {"out_channels": 3}, "sdt": {"out_channels": 1}}
    cases = []
    for model_kw, inf_head, req, allow_none in [
            (dict(heads=None, primary_head=None), None, None, True), (dict(heads=heads, primary_head=None), None, None, True),
            (dict(heads=heads, primary_head=None), None, None, False), (dict(heads=heads, primary_head="sdt"), None, None, True),
            (dict(heads=heads, primary_head="zzz"), None, None, True), (dict(heads=heads, primary_head=None), "aff", None, True),
            (dict(heads=heads, primary_head="sdt"), "aff,sdt", None, True), (dict(heads=heads, primary_head=None), "aff, sdt", None, True),
            (dict(heads=heads, primary_head=None), "aff,nope", None, True), (dict(heads=heads, primary_head=None), None, "sdt", True),
            (dict(heads=heads, primary_head=None), None, "bad", True), (dict(heads=heads, primary_head=None), None, "  ", True),
            (dict(heads={"only": {"out_channels": 2}}, primary_head=None), None, None, False)]:
        c = NS(model=NS(**model_kw), inference=NS(model=NS(head=inf_head)))
        rec = {"model": {k: v for k, v in model_kw.items()}, "inference_head": inf_head, "requested": req, "allow_none": allow_none}
        for fn_name, call in (("one", lambda: mo.resolve_output_head(c, requested_head=req, purpose="t", allow_none=allow_none)),
                              ("many", lambda: mo.resolve_output_heads(c, purpose="t"))):
            try:
                rec[fn_name] = {"value": call()}
            except Exception as e:      # noqa: BLE001
                rec[fn_name] = {"error": type(e).__name__, "message": str(e)}
        cases.append(rec)
    (HERE / "output_heads.json").write_text(json.dumps(cases, indent=0))
    print("public adapters:", len(combos), "views,", len(cases), "head cases")


def public_helpers():
    """Round 3, second pass over the public names of the in-scope modules: the reference's `resolve_output_channels` family,
    `get_total_model_head_channels`, `resolve_head_target_slice`, `normalize_channel_range_selector`,
    `infer_min_required_channels` and `validate_affinity_output` on small inputs -> tests/golden/public_helpers.json."""
    import json
    from types import SimpleNamespace as NS
    mo = S.ref("connectomics.utils.model_outputs")
    cs = S.ref("connectomics.utils.channel_slices")
    ta = S.ref("connectomics.inference.tta_affinity")

    def attempt(call):
        try:
            return {"value": call()}
        except Exception as e:      # noqa: BLE001
            return {"error": type(e).__name__, "message": str(e)}

    heads = {"aff": {"out_channels": 3, "target_slice": "0:3"}, "sdt": {"out_channels": 1}, "lsd": {"out_channels": 10, "target_slice": [4, 5]}}
    channel_cases = []
    for model_kw, inf_head, req, allow in [
            (dict(heads=None, primary_head=None, out_channels=7), None, None, True),
            (dict(heads=None, primary_head=None, out_channels=None), None, None, True),
            (dict(heads=heads, primary_head=None, out_channels=2), None, None, True),
            (dict(heads=heads, primary_head=None, out_channels=2), None, None, False),
            (dict(heads=heads, primary_head="lsd", out_channels=2), None, None, True),
            (dict(heads=heads, primary_head=None, out_channels=2), "aff,sdt", None, True),
            (dict(heads=heads, primary_head=None, out_channels=2), "aff", None, True),
            (dict(heads=heads, primary_head=None, out_channels=2), None, "sdt, lsd", True),
            (dict(heads=heads, primary_head=None, out_channels=2), None, "sdt,zzz", True),
            (dict(heads=heads, primary_head=None, out_channels=2), None, "lsd", True),
            (dict(heads=heads, primary_head=None, out_channels=2), None, "nope", True),
            (dict(heads={"only": {"out_channels": 4}}, primary_head=None, out_channels=1), None, None, False)]:
        c = NS(model=NS(**model_kw), inference=NS(model=NS(head=inf_head)))
        channel_cases.append({
            "model": dict(model_kw), "inference_head": inf_head, "requested": req, "allow": allow,
            "channels": attempt(lambda: mo.resolve_output_channels(c, requested_head=req, purpose="t", allow_ambiguous=allow)),
            "configured_channels": attempt(lambda: mo.resolve_configured_output_channels(c, purpose="t", allow_ambiguous=allow)),
            "configured_head": attempt(lambda: mo.resolve_configured_output_head(c, purpose="t", allow_none=allow)),
            "total": attempt(lambda: mo.get_total_model_head_channels(c)),
            "slices": {h: attempt(lambda h=h: mo.resolve_head_target_slice(c, h)) for h in ("aff", "sdt", "lsd", "ghost")},
            "has_inference_model": mo.get_inference_model_config(c) is not None})
    empty = NS(model=NS(heads=None))
    channel_cases.append({"model": {"heads": None}, "no_inference": True, "has_inference_model": mo.get_inference_model_config(empty) is not None,
                          "total": attempt(lambda: mo.get_total_model_head_channels(empty))})
    selectors = [None, 0, 3, -1, -4, True, "2", " -2 ", ":", "1:", ":3", "1:3", "-3:-1", ":-2", "5:2", "0:0", "1:2:3", "a:b", "", "x", [0, 2], [3, -5, "1"], [], [0.5], 1.5, (1, 4)]
    sel_cases = []
    for sel in selectors:
        shown = list(sel) if isinstance(sel, tuple) else sel
        sel_cases.append({"selector": shown, "tuple": isinstance(sel, tuple),
                          "range_form": attempt(lambda: cs.normalize_channel_range_selector(sel, context="sel")),
                          "min_channels": attempt(lambda: cs.infer_min_required_channels(sel, context="sel"))})
    plan = NS(num_channels=6, spatial_rank=3)
    aff = []
    for shape in [(1, 6, 4, 5, 5), (2, 5, 4, 5, 5), (1, 6, 5, 5), (1, 6, 2, 4, 5, 5)]:
        aff.append({"shape": list(shape), "result": attempt(lambda: ta.validate_affinity_output(plan, torch.zeros(shape)))})
    aff.append({"shape": [1, 3, 4, 4], "plan": None, "result": attempt(lambda: ta.validate_affinity_output(None, torch.zeros(1, 3, 4, 4)))})
    aff.append({"shape": [1, 6, 4, 4], "rank0": True, "result": attempt(lambda: ta.validate_affinity_output(NS(num_channels=6, spatial_rank=0), torch.zeros(1, 6, 4, 4)))})
    # channel-activation scoping of the predictor (tta.py:94-200), incl. merged-head inference ("aff,sdt": specs written against the
    # concatenated tensor, every head activated on its own slice)
    tta = S.ref("connectomics.inference.tta")
    two = {"aff": {"out_channels": 6}, "sdt": {"out_channels": 1}}
    act_cases = []
    for label, model_kw, inf_head, acts, override, width in [
            ("merged_aff", dict(heads=two, primary_head="aff", out_channels=7), "aff,sdt", [{"channels": "0:6", "activation": "scale_sigmoid"}, {"channels": "6:7", "activation": "tanh"}], "aff", 6),
            ("merged_sdt", dict(heads=two, primary_head="aff", out_channels=7), "aff,sdt", [{"channels": "0:6", "activation": "scale_sigmoid"}, {"channels": "6:7", "activation": "tanh"}], "sdt", 1),
            ("merged_straddle", dict(heads=two, primary_head="aff", out_channels=7), "aff, sdt", [{"channels": "4:", "activation": "sigmoid"}, {"channels": [0, 2], "activation": "tanh"}], "sdt", 1),
            ("merged_straddle_aff", dict(heads=two, primary_head="aff", out_channels=7), "aff, sdt", [{"channels": "4:", "activation": "sigmoid"}, {"channels": [0, 2], "activation": "tanh"}], "aff", 6),
            ("merged_request_is_list", dict(heads=two, primary_head="aff", out_channels=7), "aff,sdt", [{"channels": "0:6", "activation": "sigmoid"}], "aff,sdt", 7),
            ("head_outside_merged", dict(heads={**two, "lsd": {"out_channels": 3}}, primary_head="aff", out_channels=7), "aff,sdt", [{"channels": "0:2", "activation": "sigmoid"}], "lsd", 3),
            ("single_head_config", dict(heads=two, primary_head="sdt", out_channels=7), None, [{"channels": ":", "activation": "tanh"}], "sdt", 1),
            ("no_heads", dict(heads=None, primary_head=None, out_channels=3), None, [{"channels": "0:2", "activation": "sigmoid"}, {"channels": "2:3", "activation": "tanh"}], None, 3),
            ("overlap", dict(heads=None, primary_head=None, out_channels=3), None, [{"channels": "0:2", "activation": "sigmoid"}, {"channels": "1:3", "activation": "tanh"}], None, 3),
            ("not_a_mapping", dict(heads=None, primary_head=None, out_channels=3), None, [["0:2", "sigmoid"]], None, 3),
            ("missing_key", dict(heads=None, primary_head=None, out_channels=3), None, [{"channels": "0:2"}], None, 3),
            ("out_of_range", dict(heads=None, primary_head=None, out_channels=3), None, [{"channels": "5", "activation": "tanh"}], None, 3),
            ("none_configured", dict(heads=None, primary_head=None, out_channels=3), None, None, None, 3)]:
        c = NS(model=NS(**model_kw), inference=NS(model=NS(head=inf_head, channel_activations=acts, select_channel=None, output_dtype=None),
                                                  test_time_augmentation=NS(enabled=False)))
        built = attempt(lambda: tta.TTAPredictor(cfg=c, sliding_inferer=None, forward_fn=lambda x: x))
        rec = {"label": label, "model": dict(model_kw), "inference_head": inf_head, "activations": acts, "override": override, "width": width}
        if "error" in built:                       # the constructor parses the activations (tta.py:202-230) and refuses bad entries
            rec["construct"] = built
        else:
            pr = built["value"]
            rec["construct"] = {"value": pr.channel_activation_types}
            pr._requested_output_head_override = override
            window = attempt(lambda: pr._merged_head_window())
            if "value" in window and window["value"] is not None:
                window["value"] = list(window["value"])
            rec["window"] = window
            rec["specs"] = attempt(lambda: [[list(i), a] for i, a in pr._resolve_channel_activation_specs(width)])
        act_cases.append(rec)
    (HERE / "public_helpers.json").write_text(json.dumps({"channels": channel_cases, "selectors": sel_cases, "affinity": aff,
                                                         "activation_specs": act_cases}, indent=0))
    print("activation scoping:", len(act_cases), "cases")
    print("public helpers:", len(channel_cases), "channel cases,", len(sel_cases), "selectors,", len(aff), "affinity checks")



def comb_apply(x, flip_axes, plane, k):
    if flip_axes:
        x = torch.flip(x, dims=[a + 2 for a in flip_axes])
    if plane is not None and k % 4:
        x = torch.rot90(x, k=k, dims=[plane[0] + 2, plane[1] + 2])
    return x


# ---------------------------------------------------------------- lazy / region sliding window
def _net_lazy(x):
    ramp = torch.linspace(0, 1, x.shape[-1]).view(1, 1, 1, 1, -1)
    return torch.cat([2 * x - 1 + ramp, 0.5 * x + x.mean(dim=(2, 3, 4), keepdim=True)], 1)


def _net_ctx(x):   # reference tests/unit/test_lazy_inference.py:32-35: depends on the z-1 neighbour
    shifted = torch.zeros_like(x)
    shifted[..., 1:, :, :] = x[..., :-1, :, :]
    return x + shifted


def lazy():
    from types import SimpleNamespace as NS
    lz = S.ref("connectomics.inference.lazy")

    class FakeAccessor:
        def __init__(self, vol):
            self.vol = vol.astype(np.float32)
            self.padded_spatial_shape = tuple(vol.shape[1:])
            self.channel_count = vol.shape[0]
            self.kind = "image"

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def close(self):
            pass

        def read_patch(self, location, patch_size, *, outer_pad_mode, outer_pad_value):
            start = tuple(int(v) for v in location)
            end = tuple(start[i] + int(patch_size[i]) for i in range(3))
            shp = self.padded_spatial_shape
            lo = tuple(max(0, start[i]) for i in range(3))
            hi = tuple(min(shp[i], end[i]) for i in range(3))
            inner = self.vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
            pads = [(max(0, -start[i]), max(0, end[i] - shp[i])) for i in range(3)]
            return lz._pad_channel_first(inner, pads, mode=outer_pad_mode, constant_value=outer_pad_value)

    def cfg_for(roi, *, blending="bump", overlap=0.5, padding_mode="reflect", snap=False, ctx=(), border=(), acts=None,
                select=None, swb=3):
        return NS(
            model=NS(primary_head=None, heads=None, out_channels=2, output_size=list(roi)),
            system=NS(num_workers=0),
            data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1, use_lazy_zarr=False, use_lazy_h5=False)),
            inference=NS(
                sliding_window=NS(window_size=list(roi), sw_batch_size=swb, overlap=overlap, blending=blending,
                                  padding_mode=padding_mode, cval=0.0, keep_input_on_cpu=False, sw_device=None,
                                  output_device=None, border_mask=list(border), distributed_sharding=False,
                                  snap_to_edge=snap, target_context=list(ctx), distributed_reduce_chunk_mb=128),
                model=NS(head=None, select_channel=select, output_dtype=None, channel_activations=acts, crop_pad=None),
                test_time_augmentation=NS(enabled=False, distributed_sharding=False),
            ),
        )

    out = {}
    vols = {"arange": np.arange(4 * 5 * 6, dtype=np.float32).reshape(1, 4, 5, 6),
            "rand": np.random.default_rng(5).random((1, 20, 30, 34), dtype=np.float32)}
    for k, v in vols.items():
        out[f"vol_{k}"] = v

    def run(name, vol_key, cfg, net, region=None):
        lz._build_accessor = lambda cfg_, path, kind, mode: FakeAccessor(vols[vol_key])
        if region is None:
            y = lz.lazy_predict_volume(cfg, net, "fake://", device="cpu")
        else:
            y = lz.lazy_predict_region(cfg, net, "fake://", region_start=region[0], region_stop=region[1], device="cpu")
        out[f"{name}__y"] = y.numpy()
        print(name, tuple(y.shape), float(y.mean()))

    run("arange_identity", "arange", cfg_for((2, 3, 3)), lambda x: x)
    sig = [{"channels": "0", "activation": "sigmoid"}]
    run("rand_full", "rand", cfg_for((8, 12, 16), acts=sig), _net_lazy)
    run("rand_region", "rand", cfg_for((8, 12, 16), acts=sig), _net_lazy, region=((3, 5, 7), (17, 22, 30)))
    run("rand_snap_ctx", "rand", cfg_for((8, 12, 16), blending="distance_transform", snap=True, ctx=(1, 2, 2),
                                        border=(1, 1, 1), overlap=(0.25, 0.5, 0.5)), _net_ctx)
    run("rand_const_select", "rand", cfg_for((8, 8, 8), blending="constant", padding_mode="constant", overlap=0.0,
                                            select=[1]), _net_lazy)
    # the window grid itself (incl. face-centred boundary windows)
    for i, (img, roi, ov, snap) in enumerate([((4, 5, 6), (2, 3, 3), (0.5,) * 3, False), ((20, 30, 34), (8, 12, 16), (0.5,) * 3, False),
                                              ((20, 30, 34), (8, 12, 16), (0.25, 0.5, 0.5), True), ((165, 1024, 768), (112,) * 3, (0.5,) * 3, False)]):
        offs = lz._build_window_axis_offsets(img, roi, ov, snap_to_edge=snap)
        for a in range(3):
            out[f"grid{i}_axis{a}"] = np.asarray(offs[a], np.int64)
        out[f"grid{i}_meta"] = np.asarray(list(img) + list(roi) + [int(snap)], np.int64)
        out[f"grid{i}_ov"] = np.asarray(ov, np.float64)
    save("lazy.npz", **out)

# ---------------------------------------------------------------- weighted BCE (train-step loss)
def losses():
    """connectomics/models/losses/losses.py:17-44,190-266: WeightedBCEWithLogitsLoss values and input gradients for
    masks that are binary / broadcast over channels / real-valued with zero and negative entries / all invalid."""
    ls = S.ref("connectomics.models.losses.losses")
    g = torch.Generator().manual_seed(2024)
    out = {}
    cases = {"plain": (2, None, None), "pos_weight": (1, None, 2.5), "mask_bcast": (3, "bcast", None),
             "mask_full_pw": (2, "full", 0.4), "mask_real": (2, "real", None), "mask_none_valid": (1, "zero", None)}
    for name, (C, mk, pw) in cases.items():
        x = (torch.randn(2, C, 5, 6, 7, generator=g) * 2.5).requires_grad_()
        t = (torch.rand(2, C, 5, 6, 7, generator=g) > 0.7).float()
        w = None
        if mk == "bcast":
            w = (torch.rand(2, 1, 5, 6, 7, generator=g) > 0.3).float()
        elif mk == "full":
            w = (torch.rand(2, C, 5, 6, 7, generator=g) > 0.5).float()
        elif mk == "real":
            w = torch.randn(2, C, 5, 6, 7, generator=g)          # negative and positive weights
            w[0, 0, :2] = 0.0
        elif mk == "zero":
            w = torch.zeros(2, 1, 5, 6, 7)
        crit = ls.WeightedBCEWithLogitsLoss(pos_weight=pw)
        v = crit(x, t, weight=w)
        gx = torch.autograd.grad(v, x, allow_unused=True)[0] if v.requires_grad else None
        out[f"{name}__x"] = x.detach().numpy()
        out[f"{name}__t"] = t.numpy()
        if w is not None:
            out[f"{name}__w"] = w.numpy()
        out[f"{name}__pw"] = np.asarray([-1.0 if pw is None else pw], np.float64)
        out[f"{name}__loss"] = np.asarray([float(v)], np.float64)
        out[f"{name}__grad"] = (torch.zeros_like(x) if gx is None else gx).numpy()
        print(name, float(v))
    # regression losses (distance transforms): WeightedMSELoss / WeightedMAELoss / SmoothL1Loss, tanh option, weight maps
    for name, (cls, kw, mk) in {"mse_plain": ("WeightedMSELoss", {}, None), "mse_tanh_mask": ("WeightedMSELoss", {"tanh": True}, "bcast"),
                                "mae_real": ("WeightedMAELoss", {}, "real"), "huber_beta": ("SmoothL1Loss", {"beta": 0.3, "tanh": True}, "full"),
                                "huber_none_valid": ("SmoothL1Loss", {}, "zero")}.items():
        x = (torch.randn(2, 2, 4, 5, 6, generator=g) * 1.5).requires_grad_()
        t = torch.rand(2, 2, 4, 5, 6, generator=g) * 2 - 1
        w = None
        if mk == "bcast":
            w = (torch.rand(2, 1, 4, 5, 6, generator=g) > 0.3).float()
        elif mk == "full":
            w = (torch.rand(2, 2, 4, 5, 6, generator=g) > 0.5).float()
        elif mk == "real":
            w = torch.randn(2, 2, 4, 5, 6, generator=g)
        elif mk == "zero":
            w = torch.zeros(2, 1, 4, 5, 6)
        v = getattr(ls, cls)(**kw)(x, t, weight=w)
        gx = torch.autograd.grad(v, x, allow_unused=True)[0] if v.requires_grad else None
        out[f"reg_{name}__x"], out[f"reg_{name}__t"] = x.detach().numpy(), t.numpy()
        if w is not None:
            out[f"reg_{name}__w"] = w.numpy()
        out[f"reg_{name}__loss"] = np.asarray([float(v)], np.float64)
        out[f"reg_{name}__grad"] = (torch.zeros_like(x) if gx is None else gx).numpy()
        print(name, float(v))
    save("losses.npz", **out)


# ---------------------------------------------------------------- prediction crops (crop_pad + DeepEM affinity border)
def crops():
    import json
    from types import SimpleNamespace as NS
    aff = S.ref("connectomics.data.processing.affinity")
    cg = S.ref("connectomics.inference.chunk_grid")
    out = {"crop_pad": [], "global": [], "normalize": []}
    offset_sets = [["1-0-0", "0-1-0", "0-0-1"], ["1-0-0", "0-1-0", "0-0-1", "3-0-0", "0-9-0", "0-0-9", "0-0-27"],
                   [[-1, 0, 0], [0, -2, 0], [0, 0, 3], [2, 4, -5]], ["0-0-0"], [[4, 0, 0], [-4, 0, 0], [0, 7, -7]]]
    for offs in offset_sets:
        parsed = aff.parse_affinity_offsets(offs)
        for mode in ("deepem", "banis"):
            out["crop_pad"].append({"offsets": [list(o) for o in parsed], "mode": mode,
                                    "pad": [list(p) for p in aff.compute_affinity_crop_pad(parsed, affinity_mode=mode)]})
    for v in (None, [], [1, 2, 3], [1, 2, 3, 4, 5, 6], (0, 0, 7)):
        out["normalize"].append({"value": None if v is None else list(v), "pad": [list(p) for p in cg.normalize_crop_pad(v)]})

    def cfg_for(offsets, mode, crop_pad=None, select=None, extra_target=False):
        targets = [{"name": "affinity", "kwargs": {"offsets": offsets, "affinity_mode": mode}}]
        if extra_target:
            targets = [{"name": "binary", "kwargs": {}}] + targets
        return NS(model=NS(primary_head=None, heads=None, out_channels=len(offsets) + int(extra_target)),
                  data=NS(label_transform=NS(stack_outputs=True, targets=targets)),
                  inference=NS(model=NS(crop_pad=crop_pad, select_channel=select, head=None)))

    cases = [(offset_sets[0], "deepem", None, None, False), (offset_sets[1], "deepem", [2, 0, 0], None, False),
             (offset_sets[1], "banis", [1, 2, 3, 4, 5, 6], None, False), (offset_sets[1], "deepem", None, "0:3", False),
             (offset_sets[2], "deepem", [1, 1, 1], [0, 3], False), (offset_sets[1], "deepem", None, "1:4", True)]
    for offs, mode, cp, sel, extra in cases:
        c = cfg_for(offs, mode, cp, sel, extra)
        out["global"].append({"offsets": offs, "mode": mode, "crop_pad": cp, "select": sel, "extra_target": extra,
                              "selected_offsets": [list(o) for o in cg.resolve_selected_affinity_offsets(c)],
                              "crop": [list(p) for p in cg.resolve_global_prediction_crop(c)]})
    (HERE / "crops.json").write_text(json.dumps(out, indent=0))
    print("wrote crops.json", {k: len(v) for k, v in out.items()})


# ---------------------------------------------------------------- prediction / storage dtype transforms
def outputs():
    """connectomics/inference/output.py:146-243: apply_prediction_transform / apply_storage_dtype_transform on a float
    volume with values outside [0, 1] (clipping), for the dtype vocabulary and the scale conventions."""
    from types import SimpleNamespace as NS
    out_mod = S.ref("connectomics.inference.output")
    rng = np.random.default_rng(7)
    data = rng.uniform(-0.3, 1.4, size=(2, 4, 5, 6)).astype(np.float32)
    res = {"data": data}
    cases = {"u8_255": (255.0, "uint8"), "i8_neg_scale": (-1.0, "int8"), "u16_1000": (1000.0, "uint16"), "f16_2": (2.0, "float16"),
             "unknown_dtype": (3.0, "float8"), "scale1_none": (1.0, None), "i32_big": (1.0e6, "int32")}
    for name, (scale, dt) in cases.items():
        cfg = NS(inference=NS(prediction_transform=NS(enabled=True, intensity_scale=scale, intensity_dtype=dt), save_dtype=None))
        res["pt__" + name] = out_mod.apply_prediction_transform(cfg, data.copy())
    for dt in ("uint8", "float16", "int16", "float64"):
        res["sd__" + dt] = out_mod.apply_storage_dtype_transform(NS(inference=NS(save_dtype=dt)), data.copy() * 100.0)
    res["pt__disabled"] = out_mod.apply_prediction_transform(NS(inference=NS(prediction_transform=NS(enabled=False))), data.copy())
    save("output_transforms.npz", **res)


# ---------------------------------------------------------------- channel selectors
def selectors():
    """connectomics/utils/channel_slices.py: resolve_channel_indices / resolve_channel_range / normalize_channel_selector for
    valid and invalid selectors (errors recorded by exception type)."""
    import json
    cs = S.ref("connectomics.utils.channel_slices")
    sels = [None, 0, -1, 5, "2", "-2", ":", "0:", ":-1", "1:3", "-3:-1", "3:1", "0:0", "1:2:3", "a", "", " 1 : 4 ", [0, 2], [3, -1, 0],
            [0, 0], [], [7], (1, 2), True, 2.5, "99", "0:99", "-99:"]
    rows = []
    for sel in sels:
        for nc in (1, 3, 6):
            row = {"selector": list(sel) if isinstance(sel, tuple) else sel, "tuple": isinstance(sel, tuple), "num_channels": nc}
            for name in ("resolve_channel_indices", "resolve_channel_range"):
                try:
                    v = getattr(cs, name)(sel, num_channels=nc, context="sel")
                    row[name] = list(v)
                except Exception as e:               # noqa: BLE001 - the error type is the fixture
                    row[name] = {"error": type(e).__name__}
            rows.append(row)
    (HERE / "channel_selectors.json").write_text(json.dumps(rows))
    print("wrote channel_selectors.json", len(rows))


# ---------------------------------------------------------------- learning-rate schedule
def schedules():
    """connectomics/training/optimization/lr_scheduler.py:51-106: WarmupCosineLR learning rates per iteration for two
    parameter groups (linear warm-up, eta_min)."""
    import json
    lrs = S.ref("connectomics.training.optimization.lr_scheduler")
    rows = []
    for max_iters, wi, wf, eta in ((50, 10, 0.001, 0.0), (30, 0, 0.1, 1e-5), (40, 60, 0.01, 1e-6), (25, 5, 1.0, 0.0)):
        ps = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD([{"params": [ps[0]], "lr": 0.1}, {"params": [ps[1]], "lr": 0.02}], lr=0.1)
        sch = lrs.WarmupCosineLR(opt, max_iters=max_iters, warmup_factor=wf, warmup_iters=wi, eta_min=eta)
        seq = []
        for _ in range(max_iters):
            seq.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sch.step()
        rows.append({"max_iters": max_iters, "warmup_iters": wi, "warmup_factor": wf, "eta_min": eta, "lrs": seq})
    (HERE / "lr_schedule.json").write_text(json.dumps(rows))
    print("wrote lr_schedule.json", len(rows))


# ---------------------------------------------------------------- optimizer parameter groups
def _opt_model():
    torch.manual_seed(0)
    m = torch.nn.Sequential()
    m.add_module("conv", torch.nn.Conv3d(1, 4, 3))
    m.add_module("gn", torch.nn.GroupNorm(2, 4))
    m.add_module("act", torch.nn.PReLU())
    m.add_module("conv2", torch.nn.Conv3d(4, 4, 1, bias=False))
    m.add_module("bn", torch.nn.BatchNorm3d(4))
    m.add_module("head", torch.nn.Conv3d(4, 2, 1))
    m.add_module("tied", torch.nn.Conv3d(4, 2, 1))
    m.tied.weight = m.head.weight                 # shared parameter: listed once
    m.conv2.weight.requires_grad_(False)          # frozen: not listed
    return m


def optimizers():
    """connectomics/training/optimization/build.py:47-160: per-parameter (lr, weight_decay) and optimizer class / defaults."""
    import json
    from types import SimpleNamespace as NS
    ob = S.ref("connectomics.training.optimization.build")
    rows = []
    for oc in (dict(name="AdamW", lr=1e-3, weight_decay=0.01), dict(name="adamw", lr=2e-4, weight_decay=0.05, weight_decay_norm=0.001,
               weight_decay_bias=0.0, bias_lr_factor=2.0, betas=[0.8, 0.95], eps=1e-6),
               dict(name="Adam", lr=1e-3, weight_decay=0.0), dict(name="SGD", lr=0.1, weight_decay=1e-4, momentum=0.8)):
        m = _opt_model()
        opt = ob.build_optimizer(NS(optimization=NS(optimizer=NS(**oc))), m)
        names = {id(p): n for n, p in m.named_parameters()}
        per = {}
        for g in opt.param_groups:
            for p in g["params"]:
                per[names[id(p)]] = [g["lr"], g["weight_decay"]]
        g0 = opt.param_groups[0]
        rows.append({"cfg": oc, "class": type(opt).__name__, "per_param": per, "betas": list(g0.get("betas", [])),
                     "eps": g0.get("eps"), "momentum": g0.get("momentum")})
    (HERE / "optimizer_groups.json").write_text(json.dumps(rows))
    print("wrote optimizer_groups.json", [r["class"] for r in rows])


# ---------------------------------------------------------------- resume manifest file written by the reference
def manifests():
    """connectomics/chunked/manifest.py: a manifest file as the reference writes it (config + completed keys), so the
    on-disk format stays interchangeable (a run can be resumed by either implementation)."""
    import tempfile
    man = S.ref("connectomics.chunked.manifest")
    with tempfile.TemporaryDirectory() as d:
        path = Path(d) / "m.json"
        m = man.ResumeManifest.load_or_create(path, {"chunk_shape": [4, 5, 6], "output_shape": [10, 13, 9], "halo": [1, 2, 1], "overlap": 0})
        for k in ("z0_y0_x0", "z1_y2_x1", "z0_y0_x0", "z2_y0_x1"):
            m.mark_completed(k)
        m.mark_many(["z2_y2_x1", "z1_y2_x1"])
        (HERE / "resume_manifest_ref.json").write_text(path.read_text())
    print("wrote resume_manifest_ref.json")


# ---------------------------------------------------------------- LazyVolumeAccessor (disk-backed reader) on an HDF5 file
def accessor():
    """connectomics/inference/lazy.py:456-917: the reference's own LazyVolumeAccessor reading HDF5 files, with the in-repo
    h5lite module standing in for the absent `h5py` (both are thin layers over libhdf5; the files are ordinary HDF5) and the
    REAL `smart_normalize` / `_detect_format` of the reference (imageio / cv2, which those modules import but these functions
    do not use, are empty stubs).  Stores the input volumes and the patches / regions / shapes the reference returns."""
    import tempfile
    import types
    REPO = HERE.parent.parent
    sys.path.insert(0, str(REPO))
    from pytorch_connectomics_amd.utils import h5lite
    assert h5lite.available(), "libpytc_h5.so is needed to generate the accessor fixtures"
    S.install()
    sys.modules["h5py"] = h5lite
    for name in ("imageio", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    for name in ("connectomics.data.augmentation.augment_ops", "connectomics.data.io.io", "connectomics.inference.lazy"):
        sys.modules.pop(name, None)          # drop the stubs / an earlier stub-based import: the real modules are wanted here
    lz = S.ref("connectomics.inference.lazy")
    rng = np.random.default_rng(21)
    vols = {"zyx": (rng.random((12, 14, 18)) * 200).astype(np.uint8),                 # no channel axis
            "czyx": rng.random((2, 10, 12, 16)).astype(np.float32),                   # channel first
            "zyxc": (rng.random((10, 12, 16, 3)) * 1000).astype(np.uint16)}           # channel last
    out = {}
    cases = {
        "plain": ("zyx", dict(kind="image"), [((0, 0, 0), (6, 7, 8)), ((-2, 3, 12), (6, 8, 10)), ((8, 10, 14), (8, 8, 8))], "reflect", 0.0),
        "transpose_pad_reflect_div": ("zyx", dict(kind="image", transpose_axes=(2, 0, 1), context_pad=((2, 1), (0, 3), (2, 2)),
                                                  context_pad_mode="reflect", normalize_mode="divide-255"),
                                      [((0, 0, 0), (8, 8, 8)), ((-3, -1, 5), (10, 9, 12)), ((15, 6, 10), (8, 8, 8))], "constant", 0.25),
        "resize_bilinear_znorm": ("czyx", dict(kind="image", scale_factors=(1.5, 0.75, 1.25), context_pad=((1, 1), (1, 1), (1, 1)),
                                               context_pad_mode="constant", normalize_mode="normal", clip_percentile_low=0.05,
                                               clip_percentile_high=0.95),
                                  [((0, 0, 0), (8, 6, 10)), ((5, 2, 8), (8, 8, 8)), ((-1, -2, 14), (6, 6, 10))], "replicate", 0.0),
        "channel_last_edge_01": ("zyxc", dict(kind="image", context_pad=((0, 2), (2, 0), (1, 1)), context_pad_mode="edge",
                                              normalize_mode="0-1"),
                                 [((0, 0, 0), (6, 6, 6)), ((6, 8, 10), (6, 8, 8))], "reflect", 0.0),
        "mask_nearest_binarize": ("zyx", dict(kind="mask", scale_factors=(0.5, 2.0, 1.0), binarize=True, threshold=100.0),
                                  [((0, 0, 0), (4, 10, 8)), ((2, 20, 10), (4, 8, 8))], "constant", 0.0),
    }
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for k, v in vols.items():
            paths[k] = str(Path(d) / f"{k}.h5")
            with h5lite.File(paths[k], "w") as fh:
                fh.create_dataset("main", data=v, compression="gzip")
            out[f"vol_{k}"] = v
        for name, (vk, kw, reads, outer_mode, outer_val) in cases.items():
            with lz.LazyVolumeAccessor(paths[vk], **kw) as acc:
                out[f"{name}__shapes"] = np.asarray([acc.channel_count, *acc.raw_spatial_shape, *acc.logical_spatial_shape,
                                                     *acc.transformed_spatial_shape, *acc.padded_spatial_shape], np.int64)
                for i, (loc, size) in enumerate(reads):
                    out[f"{name}__patch{i}"] = acc.read_patch(loc, size, outer_pad_mode=outer_mode, outer_pad_value=outer_val)
                out[f"{name}__full"] = acc.load_full()
    save("lazy_accessor.npz", **out)



def masks():
    """connectomics/inference/tta.py:465-574, :1568-1617: the reference predictor's mask application (binarised mask, rank / batch /
    channel normalisation, optional centre alignment, tanh channels filled with -1) -> tests/golden/mask_application.npz + .json."""
    import json
    from types import SimpleNamespace as NS
    tta = S.ref("connectomics.inference.tta")
    sys.path.insert(0, str(HERE.parent))
    from lazy_tta_cases import mask_cases
    out, index = {}, []
    g = torch.Generator().manual_seed(29)
    for label, shape, make, align, types, apply_mask in mask_cases():
        cfg = NS(model=NS(heads=None, primary_head=None, out_channels=shape[1]),
                 inference=NS(model=NS(head=None, channel_activations=None, select_channel=None, output_dtype=None),
                              test_time_augmentation=NS(enabled=False, apply_mask=apply_mask)))
        pr = tta.TTAPredictor(cfg=cfg, sliding_inferer=None, forward_fn=lambda x: x)
        pr.channel_activation_types = types
        pred = torch.rand(*shape, generator=g) * 2 - 1
        mask = make()
        rec = {"label": label}
        try:
            res = pr._apply_mask_to_result(pred.clone(), mask, align)
            out[f"{label}__result"] = res.numpy()
        except Exception as e:      # noqa: BLE001
            rec.update(error=type(e).__name__, message=str(e))
        out[f"{label}__pred"] = pred.numpy()
        index.append(rec)
    save("mask_application.npz", **out)
    (HERE / "mask_application.json").write_text(json.dumps(index, indent=0))
    print("mask cases:", len(index), "errors:", sum("error" in r for r in index))



def _lazy_tta_cases():
    sys.path.insert(0, str(HERE.parent))
    import lazy_tta_cases as L
    return L.LAZY_TTA_CASES, L.lazy_tta_cfg


def lazy_tta():
    """connectomics/inference/lazy.py:986-1258 with test-time augmentation and a mask volume: the reference's lazy loop runs its
    TTAPredictor on every window batch (views -> forward -> inverse view -> activations / channel selection -> streaming
    ensemble -> mask), then blends.  Numpy-backed accessors as in lazy(); -> tests/golden/lazy_tta.npz."""
    LAZY_TTA_CASES, lazy_tta_cfg = _lazy_tta_cases()
    lz = S.ref("connectomics.inference.lazy")

    class FakeAccessor:
        def __init__(self, vol, kind):
            self.vol = vol.astype(np.float32)
            self.padded_spatial_shape = tuple(vol.shape[1:])
            self.channel_count = vol.shape[0]
            self.kind = kind

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def close(self):
            pass

        def read_patch(self, location, patch_size, *, outer_pad_mode, outer_pad_value):
            start = tuple(int(v) for v in location)
            end = tuple(start[i] + int(patch_size[i]) for i in range(3))
            shp = self.padded_spatial_shape
            lo = tuple(max(0, start[i]) for i in range(3))
            hi = tuple(min(shp[i], end[i]) for i in range(3))
            inner = self.vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
            pads = [(max(0, -start[i]), max(0, end[i] - shp[i])) for i in range(3)]
            return lz._pad_channel_first(inner, pads, mode=outer_pad_mode, constant_value=outer_pad_value)

    rng = np.random.default_rng(17)
    vol = rng.random((1, 20, 30, 34), dtype=np.float32)
    mask = (rng.random((1, 20, 30, 34)) > 0.35).astype(np.float32)
    out = {"vol": vol, "mask": mask}
    for name, case in LAZY_TTA_CASES.items():
        cfg = lazy_tta_cfg(**case["cfg"])
        lz._build_accessor = lambda cfg_, path, kind, mode: FakeAccessor(mask if kind == "mask" else vol, kind)
        kw = dict(mask_path="fake://mask" if case.get("mask") else None, mask_align_to_image=False, device="cpu")
        if case.get("region") is None:
            y = lz.lazy_predict_volume(cfg, _net_lazy, "fake://", **kw)
        else:
            y = lz.lazy_predict_region(cfg, _net_lazy, "fake://", region_start=case["region"][0], region_stop=case["region"][1], **kw)
        out[f"{name}__y"] = y.numpy()
        print(name, tuple(y.shape), float(y.mean()))
    save("lazy_tta.npz", **out)


# ---------------------------------------------------------------- tile-grid sources through the reference accessor
def accessor_tiles():
    """connectomics/inference/lazy.py:61-157, :675-708 + data/io/tiles.py:19-156: the reference's LazyVolumeAccessor on tile-grid
    sources (metadata JSON with relative `{row}_{column}` patterns, a missing tile, `tile_st`, `tile_ratio`, RGB label tiles, and
    a directory whose metadata is inferred).  `imageio` is absent from the image: its `imread` is provided by Pillow here (PNG is
    lossless, both decode to the same pixel values); scipy's zoom is the real one."""
    import tempfile
    import types
    from PIL import Image
    sys.path.insert(0, str(HERE.parent.parent))
    sys.path.insert(0, str(HERE.parent))
    from accessor_cases import TILE_CASES, write_tile_layout          # shared with the tests that read the fixture
    S.install()
    imageio = types.ModuleType("imageio")
    imageio.imread = lambda f: np.asarray(Image.open(f))
    imageio.v2 = types.ModuleType("imageio.v2")
    imageio.v2.imread = imageio.imread
    sys.modules["imageio"], sys.modules["imageio.v2"] = imageio, imageio.v2
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    REPO = HERE.parent.parent
    sys.path.insert(0, str(REPO))
    from pytorch_connectomics_amd.utils import h5lite
    sys.modules.setdefault("h5py", h5lite)
    for name in ("connectomics.data.augmentation.augment_ops", "connectomics.data.io.io", "connectomics.data.io.tiles",
                 "connectomics.data.io.utils", "connectomics.inference.lazy"):
        sys.modules.pop(name, None)
    lz = S.ref("connectomics.inference.lazy")
    rng = np.random.default_rng(33)
    tiles = (rng.random((4, 3, 2, 8, 10)) * 255).astype(np.uint8)
    rgb_tiles = (rng.random((4, 3, 2, 8, 10, 3)) * 255).astype(np.uint8)
    rgb_tiles[..., 0] //= 64                                    # ids below 2^24 / 64: exact in the fp32 the accessor returns
    out = {"tiles": tiles, "rgb_tiles": rgb_tiles}
    with tempfile.TemporaryDirectory() as d:
        for name, (layout, kw, reads, outer_mode, outer_val) in TILE_CASES.items():
            src = write_tile_layout(d, layout, tiles, rgb_tiles)
            with lz.LazyVolumeAccessor(src, **kw) as acc:
                assert acc.fmt == "tile"
                out[f"{name}__shapes"] = np.asarray([acc.channel_count, *acc.raw_spatial_shape, *acc.logical_spatial_shape,
                                                     *acc.transformed_spatial_shape, *acc.padded_spatial_shape], np.int64)
                for i, (loc, size) in enumerate(reads):
                    out[f"{name}__patch{i}"] = acc.read_patch(loc, size, outer_pad_mode=outer_mode, outer_pad_value=outer_val)
                out[f"{name}__full"] = acc.load_full()
                print(name, out[f"{name}__shapes"], out[f"{name}__full"].dtype, float(out[f"{name}__full"].max()))
    save("lazy_accessor_tiles.npz", **out)


# ---------------------------------------------------------------- deep-supervision loss through the reference orchestrator
def ds_loss():
    """connectomics/training/losses/orchestrator.py: LossOrchestrator.compute_deep_supervision_loss / compute_standard_loss with
    the reference's WeightedBCEWithLogitsLoss (models/losses/losses.py) on 5 output scales: logits beyond the +-20 clamp, dense
    float targets (trilinear resize + range clamp, orchestrator.py:892-952), a batch mask (nearest resize).  Stores inputs,
    totals, per-scale values and the gradients w.r.t. every output."""
    from types import SimpleNamespace as NS
    S._stub_pkg("connectomics.training.losses")
    S._stub_pkg("connectomics.config.pipeline")
    meta = S.ref("connectomics.models.losses.metadata")
    ml = sys.modules["connectomics.models.losses"]
    for n in dir(meta):
        if not n.startswith("_"):
            setattr(ml, n, getattr(meta, n))
    losses_mod = S.ref("connectomics.models.losses.losses")
    orch = S.ref("connectomics.training.losses.orchestrator")
    cfg = NS(model=NS(loss=NS(deep_supervision=True, deep_supervision_weights=[1.0, 0.5, 0.25, 0.125, 0.0625],
                              deep_supervision_clamp_min=-20.0, deep_supervision_clamp_max=20.0,
                              losses=[{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0}], loss_balancing=None),
                      primary_head=None, heads=None, out_channels=2), data=NS(label_transform=None))
    o = orch.LossOrchestrator(cfg, torch.nn.ModuleList([losses_mod.WeightedBCEWithLogitsLoss()]), [1.0], enable_nan_detection=False,
                              debug_on_nan=False, resolve_affinity_mode_fn=lambda c: None)
    g = torch.Generator().manual_seed(0)
    out = {}
    outs = {"output": (torch.randn(2, 2, 16, 16, 16, generator=g) * 12).requires_grad_(True)}
    for i, sz in enumerate((8, 4, 2, 1), 1):
        outs[f"ds_{i}"] = (torch.randn(2, 2, sz, sz, sz, generator=g) * 30).requires_grad_(True)
    lab = (torch.rand(2, 2, 16, 16, 16, generator=g) > 0.7).float()
    lab[:, 1] = torch.rand(2, 16, 16, 16, generator=g)                    # a real-valued channel in [0, 1]
    mask = (torch.rand(2, 1, 16, 16, 16, generator=g) > 0.3).float()
    for name, m in (("nomask", None), ("mask", mask)):
        for t in outs.values():
            t.grad = None
        tot, logs = o.compute_deep_supervision_loss(outs, lab, stage="train", mask=m)
        tot.backward()
        out[f"{name}__total"] = np.float32(tot.item())
        out[f"{name}__scales"] = np.asarray([logs[f"train_loss_scale_{i}"] for i in range(5)], np.float32)
        for k, t in outs.items():
            out[f"{name}__grad_{k}"] = t.grad.numpy().copy()
    # single-scale (standard) path: the clamp applies there too
    outs["output"].grad = None
    tot, _ = o.compute_standard_loss(outs["output"], lab, stage="train", mask=mask)
    tot.backward()
    out["standard__total"] = np.float32(tot.item())
    out["standard__grad_output"] = outs["output"].grad.numpy().copy()
    for k, t in outs.items():
        out[f"in_{k}"] = t.detach().numpy()
    out["label"], out["mask"] = lab.numpy(), mask.numpy()
    # the resize helper on its own: [-1, 1] targets keep their range, integer labels stay integer
    sdt = torch.rand(1, 1, 9, 10, 11, generator=g) * 2 - 1
    out["sdt"], out["sdt_resized"] = sdt.numpy(), orch.match_target_to_output(sdt, torch.zeros(1, 1, 4, 5, 5)).numpy()
    ilab = torch.randint(0, 5, (1, 1, 9, 10, 11), generator=g)
    out["ilab"], out["ilab_resized"] = ilab.numpy(), orch.match_target_to_output(ilab, torch.zeros(1, 1, 4, 5, 5)).numpy()
    save("ds_loss.npz", **out)


def loss_orchestration():
    """connectomics/training/losses/orchestrator.py:500-890 + plan.py: the training loss as the reference's LossOrchestrator assembles
    it for the term lists of tests/loss_cases.py -- class-balancing weight maps for `weight`-taking losses, term masks, pos_weight
    spellings, batch masks, deep supervision, terms skipped on the deep-supervision scales -> tests/golden/loss_orchestration.npz
    (total loss and the gradient of every output scale)."""
    import warnings
    warnings.filterwarnings("ignore")
    sys.path.insert(0, str(HERE.parent))
    from loss_cases import CASES, loss_cfg, loss_tensors
    S._stub_pkg("connectomics.training.losses")
    S._stub_pkg("connectomics.config.pipeline")
    meta = S.ref("connectomics.models.losses.metadata")
    ml = sys.modules["connectomics.models.losses"]
    for n in dir(meta):
        if not n.startswith("_"):
            setattr(ml, n, getattr(meta, n))
    ls = S.ref("connectomics.models.losses.losses")
    orch = S.ref("connectomics.training.losses.orchestrator")
    make = {"WeightedBCEWithLogitsLoss": ls.WeightedBCEWithLogitsLoss, "WeightedMSELoss": ls.WeightedMSELoss, "WeightedMAELoss": ls.WeightedMAELoss,
            "SmoothL1Loss": ls.SmoothL1Loss, "PerChannelBCEWithLogitsLoss": ls.PerChannelBCEWithLogitsLoss,
            "BCEWithLogitsLoss": torch.nn.BCEWithLogitsLoss, "MSELoss": torch.nn.MSELoss}
    out = {}
    for index, (label, terms, ds, use_mask) in enumerate(CASES):
        mods = torch.nn.ModuleList([make[t["function"]](**dict(t.get("kwargs", {}))) for t in terms])
        o = orch.LossOrchestrator(loss_cfg(terms, ds), mods, [float(t.get("coefficient", t.get("weight", 1.0))) for t in terms],
                                  enable_nan_detection=False, debug_on_nan=False, resolve_affinity_mode_fn=lambda c: None)
        outs, lab, mask = loss_tensors(index)
        outs = {k: v.requires_grad_(True) for k, v in outs.items()}
        if ds:
            total, _ = o.compute_deep_supervision_loss(outs, lab, stage="train", mask=mask if use_mask else None)
        else:
            total, _ = o.compute_standard_loss(outs["output"], lab, stage="train", mask=mask if use_mask else None)
        total.backward()
        out[f"{label}__total"] = np.float64(total.item())
        for k, v in outs.items():
            if v.grad is not None:
                out[f"{label}__grad_{k}"] = v.grad.numpy().copy()
        print(label, float(total))
    save("loss_orchestration.npz", **out)


def losses_extra():
    """models/losses/losses.py:269-351 PerChannelBCEWithLogitsLoss (per-channel class balancing, capped ratio, a channel without
    positives, valid-mask weights) values + gradients, and the orchestrator's scalar `pos_weight: auto`
    (training/losses/orchestrator.py:180-197: min(neg / pos, 10) over the valid voxels, 1 when a class is absent)."""
    from types import SimpleNamespace as NS
    ls = S.ref("connectomics.models.losses.losses")
    g = torch.Generator().manual_seed(77)
    out = {}
    for name, (C, mk, kw) in {"pc_plain": (3, None, {}), "pc_cap": (2, None, {"max_pos_weight": 1.5}), "pc_mask": (3, "full", {"max_pos_weight": 50.0}),
                              "pc_noauto": (2, "full", {"auto_pos_weight": False}), "pc_sum": (2, None, {"reduction": "sum"})}.items():
        x = (torch.randn(2, C, 4, 6, 5, generator=g) * 2).requires_grad_()
        t = (torch.rand(2, C, 4, 6, 5, generator=g) > 0.8).float()
        t[:, 0] = 0.0                                              # a channel without positives: weight stays 1
        w = (torch.rand(2, C, 4, 6, 5, generator=g) > 0.4).float() if mk else None
        v = ls.PerChannelBCEWithLogitsLoss(**kw)(x, t, weight=w)
        out[f"{name}__x"], out[f"{name}__t"] = x.detach().numpy(), t.numpy()
        if w is not None:
            out[f"{name}__w"] = w.numpy()
        out[f"{name}__loss"] = np.asarray([float(v)], np.float64)
        out[f"{name}__grad"] = torch.autograd.grad(v, x)[0].numpy()
        out[f"{name}__kw"] = np.asarray([float(kw.get("max_pos_weight", 10.0)), float(kw.get("auto_pos_weight", True)),
                                         1.0 if kw.get("reduction") == "sum" else 0.0])
        print(name, float(v))
    S._stub_pkg("connectomics.training.losses")
    S._stub_pkg("connectomics.config.pipeline")
    meta = S.ref("connectomics.models.losses.metadata")
    ml = sys.modules["connectomics.models.losses"]
    for n in dir(meta):
        if not n.startswith("_"):
            setattr(ml, n, getattr(meta, n))
    orch = S.ref("connectomics.training.losses.orchestrator")
    cfg = NS(model=NS(loss=NS(deep_supervision=False, deep_supervision_weights=[1.0], deep_supervision_clamp_min=-20.0,
                              deep_supervision_clamp_max=20.0, losses=[{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0}],
                              loss_balancing=None), primary_head=None, heads=None, out_channels=1), data=NS(label_transform=None))
    o = orch.LossOrchestrator(cfg, torch.nn.ModuleList([ls.WeightedBCEWithLogitsLoss()]), [1.0], enable_nan_detection=False,
                              debug_on_nan=False, resolve_affinity_mode_fn=lambda c: None)
    for name, (frac, masked) in {"auto_sparse": (0.97, False), "auto_mask": (0.6, True), "auto_nopos": (2.0, False)}.items():
        t = (torch.rand(2, 1, 5, 6, 7, generator=g) > frac).float()
        m = (torch.rand(2, 1, 5, 6, 7, generator=g) > 0.5).float() if masked else None
        out[f"{name}__t"] = t.numpy()
        if m is not None:
            out[f"{name}__m"] = m.numpy()
        out[f"{name}__pw"] = np.asarray([o._compute_auto_pos_weight_scalar(t, valid_mask=m)], np.float64)
        print(name, out[f"{name}__pw"])
    save("losses_extra.npz", **out)


def volume_normalisation():
    """connectomics/data/augmentation/augment_ops.py:552-611 `smart_normalize` on whole volumes -- what the reference's test pipeline
    applies to a loaded image under `data.image_transform` (default mode "0-1"); `cv2`, which the module imports and this function
    does not use, is an empty stub.  Inputs and outputs for every mode, with and without percentile clipping, on uint8 / uint16 /
    float volumes, a constant volume, and the error messages."""
    import json
    import types
    S.install()
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.modules.pop("connectomics.data.augmentation.augment_ops", None)
    ops_ = S.ref("connectomics.data.augmentation.augment_ops")
    rng = np.random.default_rng(77)
    vols = {"u8": (rng.random((6, 9, 11)) * 255).astype(np.uint8), "u16": (rng.random((5, 8, 8)) * 4000).astype(np.uint16),
            "f32": rng.normal(3.0, 2.0, (4, 7, 9)).astype(np.float32), "const": np.full((3, 4, 5), 7, np.uint8)}
    out, cases, errors = {}, [], {}
    for vn, vol in vols.items():
        out[f"in_{vn}"] = vol
        for mode in ("none", "normal", "0-1", "divide-255", "divide-2.5", "divide"):
            for lo, hi in ((0.0, 1.0), (0.02, 0.98), (0.0, 0.9)):
                key = f"{vn}__{mode}__{lo}__{hi}"
                kw = dict(divide_value=4.0) if mode == "divide" else {}
                out[key] = ops_.smart_normalize(vol, mode, clip_percentile_low=lo, clip_percentile_high=hi, **kw)
                cases.append([vn, mode, lo, hi, kw.get("divide_value")])
    for mode, kw in (("divide", {}), ("divide", dict(divide_value=0.0)), ("divide-x", {}), ("zscore", {})):
        try:
            ops_.smart_normalize(vols["u8"], mode, **kw)
        except Exception as e:          # noqa: BLE001
            errors[f"{mode}|{kw.get('divide_value')}"] = [type(e).__name__, str(e)]
    save("smart_normalize", **out)
    (HERE / "smart_normalize.json").write_text(json.dumps({"cases": cases, "errors": errors}, indent=1))


def config_defaults():
    """The reference's schema defaults (connectomics/config/schema/root.py `Config()`, plain dataclasses) as a JSON tree: what a YAML that
    does not name a key gets.  tests/test_host_config_main.py compares this package's `Config()` with it key by key."""
    import dataclasses
    import json
    S.install()
    root = S.ref("connectomics.config.schema.root")

    def plain(v):
        if dataclasses.is_dataclass(v):
            return {f.name: plain(getattr(v, f.name)) for f in dataclasses.fields(v)}
        if isinstance(v, dict):
            return {str(k): plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return v if isinstance(v, (str, int, float, bool, type(None))) else repr(v)
    (HERE / "config_defaults.json").write_text(json.dumps(plain(root.Config()), indent=1, sort_keys=True))
    print("wrote config_defaults.json")



def balancing():
    """connectomics/training/losses/balancing.py: UncertaintyLossWeighter and GradNormLossWeighter `combine` on a two-layer toy model
    with three task losses: totals, the task weights, the gradients w.r.t. the weighter's own parameters and the model, a second
    step (initial losses frozen at the first), eval mode."""
    S._stub_pkg("connectomics.training.losses")
    bal = S.ref("connectomics.training.losses.balancing")
    out = {}
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    out["w0"], out["b0"], out["w2"], out["b2"] = (p.detach().numpy().copy() for p in net.parameters())
    x = torch.randn(7, 6)
    tgt = torch.randn(7, 3)
    out["x"], out["tgt"] = x.numpy(), tgt.numpy()

    def tasks():
        y = net(x)
        return [((y[:, 0] - tgt[:, 0]) ** 2).mean() * 1.0, (y[:, 1] - tgt[:, 1]).abs().mean() * 0.5, torch.nn.functional.softplus(y[:, 2] * tgt[:, 2]).mean() * 2.0]
    names = ["a", "b", "c"]
    uw = bal.UncertaintyLossWeighter(3)
    with torch.no_grad():
        uw.log_vars.copy_(torch.tensor([0.3, -0.2, 0.0]))
    tot, wts, _ = uw.combine(tasks(), names, "train")
    tot.backward()
    out["unc_total"], out["unc_weights"] = np.float64(tot.item()), wts.numpy()
    out["unc_grad_logvars"], out["unc_grad_w2"] = uw.log_vars.grad.numpy().copy(), net[2].weight.grad.numpy().copy()
    for strat in ("last", "first", "all"):
        net.zero_grad()
        cfg_params = bal._select_shared_parameters(net, strat)
        gw = bal.GradNormLossWeighter(3, alpha=0.5, gradnorm_lambda=1.0, shared_parameters=cfg_params)
        with torch.no_grad():
            gw.task_weights.copy_(torch.tensor([1.5, 0.7, 1.0]))
        gw.train()
        tot, wts, _ = gw.combine(tasks(), names, "train")
        tot.backward()
        out[f"gn_{strat}_total"], out[f"gn_{strat}_weights"] = np.float64(tot.item()), wts.numpy()
        out[f"gn_{strat}_grad_tw"], out[f"gn_{strat}_grad_w2"] = gw.task_weights.grad.numpy().copy(), net[2].weight.grad.numpy().copy()
        if strat == "last":      # a second step after an SGD move: ratios against the FIRST step's losses
            with torch.no_grad():
                for p in net.parameters():
                    p -= 0.05 * p.grad
                gw.task_weights -= 0.1 * gw.task_weights.grad
            net.zero_grad(); gw.task_weights.grad = None
            tot2, wts2, _ = gw.combine(tasks(), names, "train")
            tot2.backward()
            out["gn_step2_total"], out["gn_step2_weights"], out["gn_step2_grad_tw"] = np.float64(tot2.item()), wts2.numpy(), gw.task_weights.grad.numpy().copy()
            gw.eval()
            tot3, _, _ = gw.combine(tasks(), names, "val")
            out["gn_eval_total"] = np.float64(tot3.item())
            with torch.no_grad():      # restore the initial weights for the other strategies
                for p, k in zip(net.parameters(), ("w0", "b0", "w2", "b2")):
                    p.copy_(torch.from_numpy(out[k]))
    save("balancing.npz", **out)


if __name__ == "__main__":
    parts = {"manifests": manifests, "optimizers": optimizers, "schedules": schedules, "selectors": selectors, "outputs": outputs, "rsunet_train": rsunet_train, "crops": crops, "losses": losses, "grids": grids, "maps": maps, "normalise": normalise, "engine": engine, "rsunets": rsunets,
             "chunks": chunks, "tta": tta, "tta_affinity": tta_affinity, "tta_zplanes": tta_zplanes, "lazy": lazy, "lazy_tta": lazy_tta, "masks": masks, "accessor": accessor, "accessor_tiles": accessor_tiles, "ds_loss": ds_loss, "loss_orchestration": loss_orchestration, "losses_extra": losses_extra, "balancing": balancing, "public_adapters": public_adapters, "public_helpers": public_helpers, "volume_normalisation": volume_normalisation, "config_defaults": config_defaults}
    chosen = [a[2:] for a in sys.argv[1:] if a.startswith("--")] or list(parts)
    for name in chosen:
        parts[name]()
