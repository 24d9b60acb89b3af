"""Generate golden fixtures from the REFERENCE's own modules (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference through tests/golden/_ref_shim.py, feeds seeded inputs through the
reference functions of the hot path and stores inputs + expected outputs as small .npz files
next to this script.  The fixtures are data; no reference source travels with them.
"""
from __future__ import annotations

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
        g = torch.Generator().manual_seed(abs(hash(name)) % (2 ** 31))
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


if __name__ == "__main__":
    grids(); maps(); normalise(); engine(); rsunets(); chunks()
