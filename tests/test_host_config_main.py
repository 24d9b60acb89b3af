"""CPU tests: YAML config surface (stage merge, _base_, overrides, alias sync) and the CLI contract."""
import textwrap
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from pytorch_connectomics_amd.config import ConfigNode, load_config, update_from_cli
from pytorch_connectomics_amd.main import binary_jaccard, main, parse_args, read_volume

YAML = """
experiment_name: unit
save_path: {save}
_base_: [base.yaml, missing_profiles.yaml]
default:
  model:
    arch: {{type: mednext}}
    mednext: {{size: S, kernel_size: 3}}
    output_size: [112, 112, 112]
  inference:
    window: {{window_size: [112, 112, 112], overlap: 0.5, blending: bump, sw_batch_size: 8}}
    test_time_augmentation: {{enabled: true, flip_axes: all}}
train:
  optimization: {{precision: "bf16-mixed", gradient_clip_val: 1.0}}
test:
  data:
    test: {{image: "random://t?shape=8,9,10"}}
  inference:
    model: {{select_channel: [0], channel_activations: [{{channels: ":", activation: sigmoid}}]}}
"""


def _write(tmp_path):
    (tmp_path / "base.yaml").write_text("default:\n  model:\n    in_channels: 1\n    out_channels: 1\n  system:\n    seed: 7\n")
    p = tmp_path / "cfg.yaml"
    p.write_text(textwrap.dedent(YAML.format(save=tmp_path / "out")))
    return p


def test_load_config_stage_merge_and_aliases(tmp_path):
    p = _write(tmp_path)
    with pytest.warns(UserWarning, match="missing_profiles.yaml"):
        cfg = load_config(p, mode="test", overrides=["inference.window.sw_batch_size=4", "model.mednext.size=B"])
    assert cfg.system.seed == 7 and cfg.model.mednext.size == "B"                      # _base_ + override
    assert cfg.inference.sliding_window.window_size == [112, 112, 112]                 # window -> sliding_window
    assert cfg.inference.sliding_window.sw_batch_size == 4
    assert cfg.inference.sliding_window.padding_mode == "reflect"                      # schema default
    assert cfg.inference.test_time_augmentation.flip_axes == "all"
    assert cfg.inference.model.select_channel == [0] and cfg.inference.select_channel == [0]
    assert cfg.optimization.precision == "16-mixed"                                    # schema default (train section not merged in test)
    assert cfg.data.test.image.startswith("random://")
    with pytest.warns(UserWarning):
        tr = load_config(p, mode="train")
    assert tr.optimization.precision == "bf16-mixed" and tr.data.test.image is None
    assert getattr(cfg.inference.sliding_window, "nope", 3) == 3 and not hasattr(cfg.model, "zzz")
    bad = tmp_path / "bad.yaml"
    bad.write_text("bogus_section: 1\n")
    with pytest.raises(ValueError, match="Unknown top-level"):
        load_config(bad)
    with pytest.raises(ValueError, match="key.sub=value"):
        update_from_cli(ConfigNode({"a": 1}), ["a"])
    bad.write_text("default:\n  inference:\n    window: {overlap: 1.5}\n")
    with pytest.raises(ValueError, match="overlap"):
        load_config(bad)


def test_cli_contract(tmp_path):
    a = parse_args(["--config", "x.yaml", "--mode", "test", "--checkpoint", "c.ckpt", "a.b=1", "c=2"])
    assert (a.config, a.mode, a.checkpoint, a.overrides, a.fast_dev_run) == ("x.yaml", "test", "c.ckpt", ["a.b=1", "c=2"], 0)
    assert parse_args(["--config", "x", "--fast-dev-run"]).fast_dev_run == 1
    with pytest.raises(SystemExit):
        parse_args(["--mode", "test"])
    p = _write(tmp_path)
    if not torch.cuda.is_available():
        with pytest.warns(UserWarning), pytest.raises(RuntimeError, match="no CPU path"):
            main(["--config", str(p), "--mode", "train"])
        with pytest.warns(UserWarning), pytest.raises(RuntimeError, match="no CPU path"):
            main(["--config", str(p), "--mode", "test"])


def test_volume_reader_and_jaccard(tmp_path):
    v = read_volume("random://x?shape=3,4,5", seed=1)
    assert v.shape == (3, 4, 5) and v.dtype == "float32"
    import numpy as np
    np.save(tmp_path / "v.npy", v)
    assert np.array_equal(read_volume(str(tmp_path / "v.npy")), v)
    with pytest.raises(ValueError, match="unsupported"):
        read_volume("x.mrc")
    pred = torch.tensor([0.9, 0.8, 0.2, 0.1])
    lab = torch.tensor([1, 0, 1, 0])
    assert binary_jaccard(pred, lab) == pytest.approx(1 / 3)
    assert binary_jaccard(torch.zeros(4), torch.zeros(4)) == 1.0


def test_profile_engine_expands_selectors_like_the_reference(tmp_path):
    """config/pipeline/profile_engine.py semantics: a selector merges its registry payload under the target with the values already
    written there winning; positional `overrides` patch the expanded list; stage sections carry their own selectors; registries
    disappear from the result; unknown names, missing registries and selectors at non-canonical paths are errors."""
    from pytorch_connectomics_amd.config import apply_profiles, load_config
    (tmp_path / "lib.yaml").write_text("""
arch_profiles:
  rsunet_small: {arch: {type: rsunet}, rsunet: {width: [8, 16], norm: group, num_groups: 4}}
  mednext_s: {arch: {type: mednext}, mednext: {size: S, kernel_size: 3}}
loss_profiles:
  bce_dice:
    losses:
      - {function: WeightedBCEWithLogitsLoss, weight: 1.0}
      - {function: DiceLoss, weight: 1.0, kwargs: {sigmoid: true}}
optimizer_profiles:
  adamw_cos: {optimizer: {name: AdamW, lr: 3.0e-4}, scheduler: {name: WarmupCosineLR, warmup_epochs: 3}}
activation_profiles:
  sig: {channel_activations: [{channels: ":", activation: sigmoid}]}
""")
    (tmp_path / "cfg.yaml").write_text("""
_base_: [lib.yaml]
default:
  model:
    arch: {profile: rsunet_small}
    rsunet: {norm: batch}                 # written explicitly: wins over the profile's `group`
    loss:
      profile: bce_dice
      overrides: {0: {pos_weight: auto}, 1: {weight: 0.5}}
  inference:
    model: {activation_profile: sig}
train:
  optimization: {profile: adamw_cos, optimizer: {lr: 1.0e-3}}
""")
    cfg = load_config(tmp_path / "cfg.yaml", mode="train")
    assert cfg.model.arch.type == "rsunet" and cfg.model.rsunet.width == [8, 16] and cfg.model.rsunet.num_groups == 4
    assert cfg.model.rsunet.norm == "batch"
    terms = [dict(t) for t in cfg.model.loss.losses]
    assert terms[0]["pos_weight"] == "auto" and terms[1]["weight"] == 0.5 and terms[1]["kwargs"]["sigmoid"] is True
    assert not hasattr(cfg.model.loss, "overrides")
    assert cfg.optimization.optimizer.lr == 1.0e-3 and cfg.optimization.scheduler.name == "WarmupCosineLR"       # explicit lr wins
    assert cfg.optimization.scheduler.warmup_epochs == 3
    test_cfg = load_config(tmp_path / "cfg.yaml", mode="test")
    assert [dict(a) for a in test_cfg.inference.model.channel_activations] == [{"channels": ":", "activation": "sigmoid"}]
    assert test_cfg.optimization.optimizer.lr != 3.0e-4                        # the train-stage selector is not applied in test mode
    raw = {"arch_profiles": {"a": {"arch": {"type": "rsunet"}}}, "default": {"model": {"arch": {"profile": "b"}}}}
    with pytest.raises(ValueError, match=r"Unknown selector 'b' at 'default.model.arch.profile'. Available profiles: \[a\]"):
        apply_profiles(raw)
    with pytest.raises(ValueError, match="requires 'arch_profiles' to be defined"):
        apply_profiles({"default": {"model": {"arch": {"profile": "a"}}}})
    with pytest.raises(ValueError, match="Non-canonical profile selector path"):
        apply_profiles({"arch_profiles": {"a": {}}, "default": {"model": {"backbone_profile": "a"}}})
    with pytest.raises(ValueError, match="Override index 3"):
        apply_profiles({"loss_profiles": {"p": {"losses": [{"function": "DiceLoss"}]}},
                        "default": {"model": {"loss": {"profile": "p", "overrides": {3: {"weight": 2.0}}}}}})
    with pytest.raises(ValueError, match="'shared' config section has been removed"):
        apply_profiles({"shared": {}})


@pytest.mark.skipif(not Path("/root/reference/tutorials").exists(), reason="reference checkout not present")
def test_reference_tutorial_configs_resolve_through_their_own_profile_library():
    """Drop-in check against the reference's data files where they lie (never copied): every training tutorial loads in train and
    test mode, its architecture profile picks the right builder, and the loss / activation profiles expand to terms this engine knows."""
    import warnings
    from pytorch_connectomics_amd.config import load_config
    from pytorch_connectomics_amd.models import build_model
    from pytorch_connectomics_amd.training.module import _LOSSES
    tut = Path("/root/reference/tutorials")
    expect = {"syn_cremi.yaml": "rsunet", "nuc_nucmm-z.yaml": "monai_unet", "minimal.yaml": "monai_unet", "fiber_linghu26.yaml": "mednext",
              "neuron_liconn_mit.yaml": "mednext", "banis.yaml": "mednext"}
    for name, arch in expect.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cfg = load_config(tut / name, mode="train")
            test_cfg = load_config(tut / name, mode="test")
            model = build_model(cfg)
        assert cfg.model.arch.type == arch, name
        assert sum(p.numel() for p in model.parameters()) > 1e5
        for term in cfg.model.loss.losses or []:
            assert dict(term)["function"] in _LOSSES, (name, dict(term)["function"])
        assert test_cfg.inference.sliding_window.window_size is None or len(test_cfg.inference.sliding_window.window_size) in (2, 3)
    cremi = load_config(tut / "syn_cremi.yaml", mode="train")
    assert list(cremi.model.rsunet.width) == [18, 36, 48, 64, 80] and cremi.model.rsunet.norm == "batch"      # explicit `batch` over the profile's `group`


@pytest.mark.parametrize("dtype,suffix", [("uint8", ".nii"), ("int16", ".nii.gz"), ("float32", ".nii.gz")])
def test_nifti_volumes_without_nibabel(tmp_path, dtype, suffix):
    """Single-file NIfTI-1 (.nii / .nii.gz): voxels in Fortran order, returned in the reference's axis convention ((X, Y, Z) ->
    (D, H, W), (X, Y, Z, C) -> (C, D, H, W); io.py:267-306), header scaling applied only when asked for, big-endian files."""
    import gzip
    import struct
    import numpy as np
    from pytorch_connectomics_amd.utils.niftilite import nifti_shape, read_nifti, write_nifti
    rng = np.random.default_rng(1)
    vol = (rng.random((5, 6, 7)) * 100).astype(dtype)                     # (D, H, W)
    p = tmp_path / f"v{suffix}"
    write_nifti(str(p), vol)
    raw = (gzip.open(p, "rb") if suffix.endswith(".gz") else open(p, "rb")).read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\x00" and struct.unpack("<8h", raw[40:56])[:4] == (3, 7, 6, 5)
    # the stored bytes are the (X, Y, Z) array in Fortran order: x fastest
    stored = np.frombuffer(raw, dtype=np.dtype(dtype).newbyteorder("<"), offset=352).reshape((7, 6, 5), order="F")
    assert np.array_equal(stored.transpose(2, 1, 0), vol)
    got = read_nifti(str(p))
    assert got.dtype == np.dtype(dtype) and got.flags.c_contiguous and np.array_equal(got, vol)
    assert nifti_shape(str(p)) == (5, 6, 7) and np.array_equal(read_volume(str(p)), vol)
    vol4 = (rng.random((2, 3, 4, 5)) * 50).astype(dtype)                  # (C, D, H, W)
    q = tmp_path / f"c{suffix}"
    write_nifti(str(q), vol4)
    assert nifti_shape(str(q)) == (2, 3, 4, 5) and np.array_equal(read_nifti(str(q)), vol4)
    # scl_slope / scl_inter: applied like nibabel's dataobj (float64 result); a big-endian copy reads the same
    hdr = bytearray(raw[:352])
    struct.pack_into("<2f", hdr, 112, 0.5, 3.0)
    s = tmp_path / "scaled.nii"
    s.write_bytes(bytes(hdr) + raw[352:])
    sc = read_nifti(str(s))
    assert sc.dtype == np.float64 and np.allclose(sc, vol.astype(np.float64) * 0.5 + 3.0)
    be = bytearray(348)
    struct.pack_into(">i", be, 0, 348)
    struct.pack_into(">8h", be, 40, 3, 7, 6, 5, 1, 1, 1, 1)
    struct.pack_into(">h", be, 70, struct.unpack("<h", raw[70:72])[0])
    struct.pack_into(">f", be, 108, 352.0)
    struct.pack_into(">2f", be, 112, 1.0, 0.0)
    be[344:348] = b"n+1\x00"
    b = tmp_path / "big.nii"
    b.write_bytes(bytes(be) + b"\x00" * 4 + vol.transpose(2, 1, 0).astype(np.dtype(dtype).newbyteorder(">")).tobytes(order="F"))
    assert np.array_equal(read_nifti(str(b)), vol)
    (tmp_path / "bad.nii").write_bytes(b"\x00" * 400)
    with pytest.raises(ValueError, match="not a NIfTI-1"):
        read_nifti(str(tmp_path / "bad.nii"))


def test_png_slice_stack(tmp_path):
    import numpy as np
    from PIL import Image
    vol = (np.random.default_rng(2).random((4, 9, 11)) * 255).astype(np.uint8)
    for z in range(4):
        Image.fromarray(vol[z]).save(tmp_path / f"s_{z:03d}.png")
    got = read_volume(str(tmp_path / "s_*.png"))
    assert got.dtype == np.uint8 and np.array_equal(got, vol)
    rgb = (np.random.default_rng(3).random((2, 5, 6, 3)) * 255).astype(np.uint8)
    for z in range(2):
        Image.fromarray(rgb[z]).save(tmp_path / f"c_{z}.png")
    assert np.array_equal(read_volume(str(tmp_path / "c_*.png")), rgb.transpose(3, 0, 1, 2))
    with pytest.raises(ValueError, match="No files found"):
        read_volume(str(tmp_path / "none_*.png"))


def test_zarr_v2_volume(tmp_path):
    import json
    import zlib
    import numpy as np
    vol = (np.random.default_rng(4).random((5, 6, 7)) * 1000).astype(np.uint16)
    root = tmp_path / "v.zarr" / "img"
    root.mkdir(parents=True)
    (root / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": [5, 6, 7], "chunks": [3, 4, 7], "dtype": "<u2", "order": "C",
                                              "compressor": {"id": "zlib", "level": 1}, "fill_value": 0, "filters": None}))
    for iz in range(2):
        for iy in range(2):
            block = np.zeros((3, 4, 7), np.uint16)
            sub = vol[iz * 3:(iz + 1) * 3, iy * 4:(iy + 1) * 4]
            block[:sub.shape[0], :sub.shape[1]] = sub
            (root / f"{iz}.{iy}.0").write_bytes(zlib.compress(block.tobytes()))
    assert np.array_equal(read_volume(str(tmp_path / "v.zarr" / "img")), vol)
    assert np.array_equal(read_volume(str(tmp_path / "v.zarr")), vol)            # first array of the group


def test_schema_defaults_against_the_reference_schema(golden_dir):
    """Every key this package's `Config()` carries has the reference schema's default (tests/golden/config_defaults.json, generated
    from connectomics/config/schema by make_golden.py --config_defaults) -- a YAML that does not name a key must behave the same on
    both sides -- except the deviations listed here, each on purpose."""
    import json
    from pytorch_connectomics_amd.config import Config
    reference = json.loads((golden_dir / "config_defaults.json").read_text())
    on_purpose = {
        "experiment_name", "save_path",                       # naming of this package's own output directory
        "model.arch.type",                                    # reference default monai_basic_unet3d is outside the hot path
        "model.input_size", "model.output_size", "data.dataloader.patch_size",   # [128]*3 there; here a configuration names its sizes
        "model.heads", "inference.model.channel_activations", "evaluation.metrics",   # None / {} / []: the same meaning
        "model.loss.deep_supervision_weights",                # None there = the same [1, .5, .25, .125, .0625] spelled out
    }
    only_here = {"data.data_transform.patch_size", "inference.window.edge_offset", "inference.window.min_contact",
                 "inference.sliding_window.edge_offset", "inference.sliding_window.min_contact", "inference.save"}
    differing, unknown = [], []

    def plain(v):
        if hasattr(v, "items"):
            return {k: plain(x) for k, x in v.items()}
        return [plain(x) for x in v] if isinstance(v, (list, tuple)) else v

    def walk(ours, theirs, path):
        for key, value in ours.items():
            here = f"{path}.{key}" if path else key
            if not isinstance(theirs, dict) or key not in theirs:
                unknown.append(here)
            elif isinstance(value, dict) and isinstance(theirs[key], dict):
                walk(value, theirs[key], here)
            elif value != theirs[key]:
                differing.append(here)
    walk(plain(Config()), reference, "")
    assert set(differing) == on_purpose, sorted(set(differing) ^ on_purpose)
    assert set(unknown) == only_here, sorted(set(unknown) ^ only_here)
    cfg = Config()
    assert cfg.data.image_transform.normalize == "0-1" and cfg.data.dataloader.batch_size == 4 and cfg.system.num_workers == 8


def test_whole_volume_normalisation_matches_the_reference(golden_dir):
    """`normalize_volume` == the reference's `smart_normalize` (augment_ops.py:552-611) bit for bit, dtype included, on 72 mode / clip /
    dtype combinations; same error messages; `normalize_image_for_config` reads data.image_transform and leaves arrays alone when a
    configuration has no such section."""
    import json
    from pytorch_connectomics_amd.utils.volume_normalize import normalize_image_for_config, normalize_volume
    g, meta = np.load(golden_dir / "smart_normalize.npz"), json.loads((golden_dir / "smart_normalize.json").read_text())
    for vol, mode, low, high, divisor in meta["cases"]:
        want = g[f"{vol}__{mode}__{low}__{high}"]
        got = normalize_volume(g[f"in_{vol}"], mode, divide_value=divisor, clip_percentile_low=low, clip_percentile_high=high)
        assert got.dtype == want.dtype and np.array_equal(got, want), (vol, mode, low, high)
    for key, (kind, message) in meta["errors"].items():
        mode, divisor = key.split("|")
        with pytest.raises(ValueError) as err:
            normalize_volume(g["in_u8"], mode, divide_value=None if divisor == "None" else float(divisor))
        assert type(err.value).__name__ == kind and str(err.value) == message
    raw = g["in_u8"]
    assert normalize_image_for_config(raw, NS()) is raw
    assert normalize_image_for_config(raw, NS(data=NS(image_transform=NS(normalize="none")))) is raw
    scaled = normalize_image_for_config(raw, NS(data=NS(image_transform=NS(normalize="0-1", clip_percentile_low=0.0, clip_percentile_high=1.0))))
    assert np.array_equal(scaled, g["u8__0-1__0.0__1.0"]) and raw.dtype == np.uint8       # the input is not modified


def test_prepare_test_image_follows_the_reference_test_transforms():
    """val_transpose -> context border (pad_size, pad_mode) -> normalisation over the padded volume (reference
    data/augmentation/build.py:416-655); the SNEMI tutorial's pad_size [16, 80, 80] + crop_pad [15, 16, 79, 80, 79, 80] + the deepem
    affinity crop bring the prediction back to the label's field of view."""
    from pytorch_connectomics_amd.inference.crop import cropped_shape, resolve_global_prediction_crop
    from pytorch_connectomics_amd.utils.volume_normalize import normalize_volume, prepare_test_image
    rng = np.random.default_rng(3)
    vol = (rng.random((5, 6, 7)) * 200).astype(np.uint8)
    cfg = NS(data=NS(data_transform=NS(val_transpose=[2, 0, 1], pad_size=[1, 2, 3], pad_mode="reflect", resize=None),
                     image_transform=NS(normalize="normal", clip_percentile_low=0.0, clip_percentile_high=1.0)))
    want = normalize_volume(np.pad(np.transpose(vol, (2, 0, 1)), [(1, 1), (2, 2), (3, 3)], mode="reflect"), "normal")
    got = prepare_test_image(vol, cfg)
    assert got.shape == (9, 9, 12) and np.array_equal(got, want)
    four = np.stack([vol, vol[::-1]])                                   # (C, Z, Y, X): the channel axis is left alone
    cfg.data.data_transform = NS(val_transpose=[], pad_size=[1, 0, 2, 0, 0, 3], pad_mode="constant", resize=None)
    cfg.data.image_transform = NS(normalize="none")
    padded = prepare_test_image(four, cfg)
    assert padded.shape == (2, 6, 8, 10) and np.array_equal(padded[:, 1:, 2:, :7], four) and padded[:, 0].max() == 0
    cfg.data.data_transform = NS(val_transpose=None, pad_size=[0, 0, 0], pad_mode="reflect", resize=None)
    assert prepare_test_image(vol, cfg) is vol
    cfg.data.data_transform.resize = [8, 8, 8]
    with pytest.raises(NotImplementedError, match="lazy reader"):
        prepare_test_image(vol, cfg)
    snemi = NS(data=NS(label_transform=NS(stack_outputs=True, targets=[{"name": "affinity", "kwargs": {
                   "offsets": ["1-0-0", "0-1-0", "0-0-1"], "affinity_mode": "deepem"}}])),
               inference=NS(crop_pad=None, model=NS(select_channel=[0, 1, 2], crop_pad=[15, 16, 79, 80, 79, 80])))
    crop = resolve_global_prediction_crop(snemi)
    assert cropped_shape((100 + 32, 1024 + 160, 1024 + 160), crop) == (100, 1024, 1024)


def test_prepare_test_mask_and_alignment_switch():
    """A mask volume gets the image's transpose, strict `mask > threshold` binarisation under mask_transform.binarize (dtype kept) and a
    ZERO context border (reference data/augmentation/build.py:566-615); `align_to_image` comes from mask_transform, else data_transform
    (test_pipeline.py:282-288)."""
    from pytorch_connectomics_amd.utils.volume_normalize import mask_align_to_image, prepare_test_mask
    mask = np.array([[[0, 3], [255, 1]], [[2, 0], [0, 9]]], dtype=np.uint8)
    cfg = NS(data=NS(data_transform=NS(val_transpose=[0, 2, 1], pad_size=[1, 0, 1], pad_mode="reflect", align_to_image=True),
                     mask_transform=NS(binarize=True, threshold=2.0)))
    out = prepare_test_mask(mask, cfg)
    assert out.dtype == np.uint8 and out.shape == (4, 2, 4)
    assert np.array_equal(out[1:3, :, 1:3], (np.transpose(mask, (0, 2, 1)) > 2).astype(np.uint8))
    assert out[0].max() == 0 and out[-1].max() == 0 and out[:, :, 0].max() == 0 and out[:, :, -1].max() == 0
    assert mask_align_to_image(cfg) is False                       # mask_transform wins and has no align_to_image
    cfg.data.mask_transform = None
    assert mask_align_to_image(cfg) is True
    assert np.array_equal(prepare_test_mask(mask, NS(data=NS(data_transform=None, mask_transform=None))), mask)
    assert mask_align_to_image(NS()) is False
