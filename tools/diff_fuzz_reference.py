"""Differential fuzzing of the HOST logic of the hot path against the reference's own functions (build container only: imports
/root/reference through tests/golden/_ref_shim.py; nothing is copied).  Thousands of generated configurations / shapes per function
pair; a pair agrees when both return equal values or both raise the same exception type with the same message.

    python tools/diff_fuzz_reference.py            # prints one line per function pair + the first mismatches

Covers: sliding-window config resolvers, scan interval / patch grid, lazy window-grid builders, TTA view enumeration and ensemble
mode maps, chunk grid + halo regions, prediction crops, channel selectors."""
from __future__ import annotations

import itertools
import random
import sys
from pathlib import Path
from types import SimpleNamespace as NS

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests" / "golden"))
sys.path.insert(0, str(ROOT))
import _ref_shim as S  # noqa: E402


def outcome(fn):
    try:
        return ("ok", repr(_plain(fn())))
    except Exception as e:      # noqa: BLE001
        return ("err", type(e).__name__, str(e))


def _plain(v):
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, slice):
        return ("slice", v.start, v.stop, v.step)
    if hasattr(v, "__dataclass_fields__"):
        return {k: _plain(getattr(v, k)) for k in v.__dataclass_fields__}
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if hasattr(v, "tolist"):
        return v.tolist()
    return v


def stricter_on_purpose(name, ref, ours):
    """The two places where this package validates input the reference lets through: a chunk grid with a non-positive chunk extent
    (reference: silently EMPTY grid) and `inference.chunking.chunk_size` that is unset / too short / non-positive (reference: an
    accidental TypeError / IndexError, or a non-positive chunk shape that fails later).  Ours raises a ValueError that names the key."""
    if ours[0] != "err" or ours[1] != "ValueError":
        return False
    if name == "build_chunk_grid":
        return (ref == ("ok", "[]") or (ref[0] == "err" and ref[1] == "ZeroDivisionError")) and "must be positive" in ours[2]
    if name == "resolve_chunk_shape":
        accidental = ref[0] == "err" and (ref[1] in ("TypeError", "IndexError") or "chunk_size must be set" in ours[2])
        nonsense = ref[0] == "ok" and any(v <= 0 for v in eval(ref[1]))
        return accidental or nonsense
    return False


class Tally:
    def __init__(self):
        self.rows = []

    def run(self, name, cases, ref_fn, our_fn, show=3):
        n = bad = strict = 0
        first = []
        for case in cases:
            a, b = outcome(lambda: ref_fn(*case)), outcome(lambda: our_fn(*case))
            n += 1
            if a != b:
                if stricter_on_purpose(name, a, b):
                    strict += 1
                    continue
                bad += 1
                if len(first) < show:
                    first.append((case, a, b))
        self.rows.append((name, n, bad))
        print(f"{name:55s} cases {n:6d}  mismatches {bad}" + (f"  (+{strict} stricter on purpose)" if strict else ""))
        for case, a, b in first:
            print("    case", str(case)[:300], "\\n      ref ", str(a)[:300], "\\n      ours", str(b)[:300])


def main():
    rnd = random.Random(1234)
    t = Tally()
    rw, ow = S.ref("connectomics.inference.window"), __import__("pytorch_connectomics_amd.inference.window", fromlist=["x"])
    shapes = [tuple(rnd.randint(1, 70) for _ in range(3)) for _ in range(400)]
    rois = [tuple(rnd.randint(1, 40) for _ in range(3)) for _ in range(400)]
    overlaps = [rnd.choice([0.0, 0.25, 0.5, 0.75, 0.9, (0.25, 0.5, 0.5), (0.0, 0.5, 0.75)]) for _ in range(400)]
    t.run("compute_scan_interval", list(zip(shapes, rois, overlaps)),
          lambda s, r, o: rw.compute_scan_interval(tuple(max(a, b) for a, b in zip(s, r)), r, overlap=o),
          lambda s, r, o: ow.compute_scan_interval(tuple(max(a, b) for a, b in zip(s, r)), r, overlap=o))

    def grid(mod):
        def f(s, r, o):
            img = tuple(max(a, b) for a, b in zip(s, r))
            return mod.dense_patch_slices(img, r, mod.compute_scan_interval(img, r, overlap=o), return_slice=False)
        return f
    t.run("dense_patch_slices", list(zip(shapes[:200], rois[:200], overlaps[:200])), grid(rw), grid(ow))

    rl, ol = S.ref("connectomics.inference.lazy"), __import__("pytorch_connectomics_amd.inference.lazy", fromlist=["x"])
    ov3 = [o if isinstance(o, tuple) else (o,) * 3 for o in overlaps]
    snaps = [rnd.random() < 0.5 for _ in range(400)]
    big = [tuple(max(a, b) + rnd.randint(0, 30) for a, b in zip(s, r)) for s, r in zip(shapes, rois)]
    t.run("lazy._build_window_axis_offsets", list(zip(big, rois, ov3, snaps)),
          lambda s, r, o, sn: rl._build_window_axis_offsets(s, r, o, snap_to_edge=sn),
          lambda s, r, o, sn: ol._build_window_axis_offsets(s, r, o, snap_to_edge=sn))

    def region_case():
        for s, r, o, sn in zip(big[:250], rois[:250], ov3[:250], snaps[:250]):
            lo = tuple(rnd.randint(0, max(0, d - 1)) for d in s)
            hi = tuple(rnd.randint(l + 1, d) for l, d in zip(lo, s))
            yield s, r, o, lo, hi, sn
    t.run("lazy._build_intersecting_window_slices", list(region_case()),
          lambda s, r, o, lo, hi, sn: rl._build_intersecting_window_slices(s, r, o, region_start=lo, region_stop=hi, snap_to_edge=sn),
          lambda s, r, o, lo, hi, sn: ol._build_intersecting_window_slices(s, r, o, region_start=lo, region_stop=hi, snap_to_edge=sn))
    ctxs = [None, [], [1, 2, 2], [0, 0, 0], [40, 1, 1], [1, 2], "x", 3, [1.5, 2, 2], [-1, 0, 0]]
    t.run("lazy._resolve_target_context", [(NS(target_context=c), r) for c in ctxs for r in rois[:20]],
          rl._resolve_target_context, ol._resolve_target_context)

    rc, oc = S.ref("connectomics.inference.tta_combinations"), __import__("pytorch_connectomics_amd.inference.tta_combinations", fromlist=["x"])
    flips = [None, "all", "none", [], [[0]], [[0], [1, 2]], [0, 1], [[0, 1, 2]], [[3]], [[-1]], "z", [["a"]], [[0], [0]], [[1, 0]], 1, [[0, 0]]]
    rots = [None, [], [[1, 2]], [[0, 1], [1, 2]], "all", [[2, 1]], [[1, 1]], [[0, 3]], [1, 2], [[1, 2], [1, 2]], [["y", "x"]]]
    ks = [None, [], [1], [0, 1, 2, 3], [1, 3], [4], [5, -1], 2, ["a"]]
    combos = [(NS(flip_axes=f, rotation90_axes=r, rotate90_k=k), d) for f in flips for r in rots for k in ks for d in (2, 3)]
    t.run("resolve_tta_augmentation_combinations", combos, lambda c, d: rc.resolve_tta_augmentation_combinations(c, spatial_dims=d),
          lambda c, d: oc.resolve_tta_augmentation_combinations(c, spatial_dims=d))
    modes = ["mean", "min", "max", "median", None, [["0", "min"]], [["0:2", "min"], ["2", "max"]], [[":", "mean"]], [["5", "min"]],
             [["0", "min"], ["0", "max"]], [["0", "avg"]], [[0, "min"]], {"0": "min"}, [["0:2"]], [], "MEAN", [["1:", "max"], ["0", "min"]]]
    t.run("_resolve_ensemble_mode_map", [(m, c) for m in modes for c in (1, 2, 3, 6)], rc._resolve_ensemble_mode_map, oc._resolve_ensemble_mode_map)

    rg, og = S.ref("connectomics.chunked.chunk_grid"), __import__("pytorch_connectomics_amd.chunked.chunk_grid", fromlist=["x"])
    rh, oh = S.ref("connectomics.chunked.halo"), __import__("pytorch_connectomics_amd.chunked.halo", fromlist=["x"])
    vols = [tuple(rnd.randint(1, 90) for _ in range(3)) for _ in range(150)] + [(0, 4, 4), (4, 4), (5, 5, 5, 5)]
    chks = [tuple(rnd.randint(-1, 50) for _ in range(3)) for _ in range(150)] + [(2, 2, 2), (2, 2), (1, 1, 1)]
    t.run("build_chunk_grid", list(zip(vols, chks)), rg.build_chunk_grid, og.build_chunk_grid)

    def halo_cases():
        for v, c in zip(vols[:120], chks[:120]):
            c = tuple(max(1, x) for x in c)
            crop = tuple(rnd.randint(0, 3) for _ in range(3))
            inp = tuple(a + b + rnd.randint(0, 3) for a, b in zip(v, crop))
            halo = tuple(rnd.randint(0, 9) for _ in range(3))
            for which in (0, -1):
                yield v, c, inp, halo, crop, which
    t.run("resolve_halo_region", list(halo_cases()),
          lambda v, c, inp, h, cr, w: rh.resolve_halo_region(rg.build_chunk_grid(v, c)[w], inp, halo=h, crop_before=cr),
          lambda v, c, inp, h, cr, w: oh.resolve_halo_region(og.build_chunk_grid(v, c)[w], inp, halo=h, crop_before=cr))

    rk, ok_ = S.ref("connectomics.inference.chunk_grid"), __import__("pytorch_connectomics_amd.inference.chunk_grid", fromlist=["x"])
    pads = [None, 0, 3, [1, 2, 3], [1, 2, 3, 4, 5, 6], [[1, 2], [3, 4], [5, 6]], [1, 2], "3", [[1, 2], [3, 4]], -1, [1.5, 2, 3], [[1], [2], [3]], (2, 2, 2)]
    t.run("normalize_crop_pad", [(p,) for p in pads], rk.normalize_crop_pad, ok_.normalize_crop_pad)
    sizes = [None, [], [8, 8, 8], [100, 4, 4], [0, 4, 4], [4, 4], "8", [8, 8, 8, 8], [-1, 2, 2]]
    t.run("resolve_chunk_shape", [(NS(inference=NS(chunking=NS(chunk_size=s, axes=a))), f) for s in sizes for a in ("all", "z", "y", "ALL")
                                  for f in ((20, 30, 40), (5, 5, 5))], rk.resolve_chunk_shape, ok_.resolve_chunk_shape)
    t.run("resolve_h5_spatial_chunks", [(s,) for s in vols[:60]], rk.resolve_h5_spatial_chunks, ok_.resolve_h5_spatial_chunks)

    rs, os_ = S.ref("connectomics.utils.channel_slices"), __import__("pytorch_connectomics_amd.utils.channel_slices", fromlist=["x"])
    sels = [None, 0, 3, -1, -9, True, "2", " -2 ", ":", "1:", ":3", "1:3", "-3:-1", ":-2", "5:2", "0:0", "1:2:3", "a:b", "", "x", [0, 2], [3, -5, "1"],
            [], [0.5], 1.5, (1, 4), "7", "-7:", [0, 0], ["a"], "1 : 3", "+1", b"1"]
    for fn in ("normalize_channel_selector", "normalize_channel_range_selector", "infer_min_required_channels"):
        t.run(f"channel_slices.{fn}", [(s,) for s in sels], lambda s, fn=fn: getattr(rs, fn)(s, context="c"), lambda s, fn=fn: getattr(os_, fn)(s, context="c"))
    for fn in ("resolve_channel_indices", "resolve_channel_range"):
        t.run(f"channel_slices.{fn}", [(s, n) for s in sels for n in (0, 1, 3, 7)],
              lambda s, n, fn=fn: getattr(rs, fn)(s, num_channels=n, context="c"), lambda s, n, fn=fn: getattr(os_, fn)(s, num_channels=n, context="c"))
    total, bad = sum(r[1] for r in t.rows), sum(r[2] for r in t.rows)
    print(f"TOTAL {total} cases, {bad} mismatches over {len(t.rows)} function pairs")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
