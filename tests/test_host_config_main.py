"""CPU tests: YAML config surface (stage merge, _base_, overrides, alias sync) and the CLI contract."""
import textwrap

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
        read_volume("x.tiff")
    pred = torch.tensor([0.9, 0.8, 0.2, 0.1])
    lab = torch.tensor([1, 0, 1, 0])
    assert binary_jaccard(pred, lab) == pytest.approx(1 / 3)
    assert binary_jaccard(torch.zeros(4), torch.zeros(4)) == 1.0
