"""Config surface of the hot path -- a small PyYAML loader for the keys SURVEY.md section 5.6 lists
(the reference uses OmegaConf dataclasses, config/pipeline/config_io.py:264-297; neither OmegaConf nor the
reference's profile YAMLs are available to this engine, so profiles are not expanded).

Supported: `_base_` inheritance (relative paths, missing bases are skipped with a warning), the
`default` + `train|test|tune` stage sections merged into the runtime tree (stage_resolver.py:336),
`key.sub=value` CLI overrides (config_io.py:351), schema defaults for the sections the engine reads, and the
canonical `inference.window` -> runtime `inference.sliding_window` alias sync (schema/inference.py:278-331).
"""
from __future__ import annotations

import copy
import logging
import warnings
from pathlib import Path
from typing import Any, Iterable, Mapping

import yaml

logger = logging.getLogger(__name__)

_TOP_LEVEL = {"experiment_name", "description", "save_path", "_base_", "default", "train", "test", "tune",
              "tune_test", "system", "model", "data", "optimization", "monitor", "inference", "evaluation",
              "decoding", "pipeline_profile", "profile", "tuning"}

_WINDOW_KEYS = ["enabled", "window_size", "sw_batch_size", "overlap", "blending", "padding_mode", "cval",
                "keep_input_on_cpu", "distributed_sharding", "distributed_reduce_chunk_mb", "sw_device",
                "output_device", "snap_to_edge", "target_context", "border_mask"]


def _window_defaults() -> dict:
    return dict(enabled=False, window_size=None, sw_batch_size=None, overlap=0.5, blending="bump",
                padding_mode="reflect", cval=0.0, keep_input_on_cpu=False, distributed_sharding=False,
                distributed_reduce_chunk_mb=128, sw_device=None, output_device=None, snap_to_edge=False,
                target_context=[], border_mask=[], edge_offset=None, min_contact=1)


def schema_defaults() -> dict:
    """Defaults of the reference dataclasses for the sections the engine reads
    (schema/model.py:23-120, schema/model_mednext.py:7-21, schema/model_rsunet.py:7-19, schema/model_monai.py:7-22,
    schema/inference.py:21-174, schema/optimization.py, schema/system.py)."""
    return {
        "experiment_name": "experiment", "description": "", "save_path": "outputs/experiment",
        "system": {"num_gpus": 1, "num_workers": 0, "seed": 42, "accelerator": "auto"},
        "model": {
            "arch": {"type": "mednext"}, "in_channels": 1, "out_channels": 1, "input_size": None,
            "output_size": None, "heads": None, "primary_head": None,
            "mednext": {"size": "S", "kernel_size": 3, "base_channels": 32, "exp_r": 4,
                        "block_counts": [2] * 9, "do_res": True, "do_res_up_down": True,
                        "checkpoint_style": None, "norm": "group", "dim": "3d", "grn": False},
            "rsunet": {"width": [16, 32, 64, 128], "norm": "batch", "activation": "relu", "num_groups": 8,
                       "down_factors": None, "depth_2d": 0, "kernel_2d": [1, 3, 3], "act_negative_slope": 0.01,
                       "act_init": 0.25},
            "monai": {"filters": [32, 64, 128, 256, 512], "dropout": 0.0, "norm": "batch", "num_groups": 8,
                      "activation": "relu", "spatial_dims": 3, "num_res_units": 2, "kernel_size": 3, "strides": None,
                      "upsample_mode": "deconv", "upsample_interp_mode": "linear", "upsample_align_corners": True},
            "loss": {"deep_supervision": False, "deep_supervision_weights": [1.0, 0.5, 0.25, 0.125, 0.0625],
                     "deep_supervision_clamp_min": -20.0, "deep_supervision_clamp_max": 20.0, "losses": None},
        },
        "data": {"train": {"image": None, "label": None, "do_2d": False},
                 "val": {"image": None, "label": None, "do_2d": False},
                 "test": {"image": None, "label": None},
                 "dataloader": {"batch_size": 1, "patch_size": None, "use_lazy_zarr": False, "use_lazy_h5": False},
                 "data_transform": {"patch_size": None},
                 "image_transform": {"normalize": "none"}},
        # schema/optimization.py:8-113 (OptimizerConfig, SchedulerConfig, EMAConfig, OptimizationConfig)
        "optimization": {"precision": "16-mixed", "gradient_clip_val": 1.0, "accumulate_grad_batches": 1,
                         "max_epochs": 200, "max_steps": None, "n_steps_per_epoch": -1,
                         "optimizer": {"name": "AdamW", "lr": 1e-3, "weight_decay": 0.01, "momentum": 0.9,
                                       "betas": [0.9, 0.999], "eps": 1e-8},
                         "scheduler": {"profile": None, "name": "CosineAnnealingLR", "params": {}, "monitor": None,
                                       "mode": "min", "factor": 0.1, "patience": 10, "threshold": 1e-4, "cooldown": 0,
                                       "eps": 1e-8, "warmup_epochs": 10, "warmup_start_lr": 1e-4, "min_lr": 1e-5,
                                       "interval": "epoch", "frequency": 1},
                         "ema": {"enabled": False, "decay": 0.999, "warmup_steps": 0, "validate_with_ema": True,
                                 "device": None, "copy_buffers": True}},
        "monitor": {},
        "inference": {
            "model": {"head": None, "select_channel": None, "output_dtype": None, "channel_activations": None,
                      "crop_pad": None},
            "execution": {"strategy": "whole_volume", "do_eval": True},
            "window": _window_defaults(),
            "sliding_window": _window_defaults(),
            "chunking": {"enabled": False, "output_mode": "decoded", "chunk_size": None, "halo": [0, 0, 0],
                         "axes": "all", "roi": None, "shard_id": None, "num_shards": None, "temp_dir": "",
                         "save_intermediate": False},
            "test_time_augmentation": {"enabled": False, "distributed_sharding": True, "flip_axes": "all",
                                       "rotation90_axes": None, "rotate90_k": None, "patch_first_local": True,
                                       "apply_mask": True, "ensemble_mode": "mean", "empty_cache_interval": 4},
            "save": {"enabled": True, "format": "h5"},
            "prediction_transform": {"enabled": False, "intensity_scale": -1.0, "intensity_dtype": None},
            "save_dtype": None,
        },
        "evaluation": {"enabled": False, "metrics": []},
    }


class ConfigNode(dict):
    """dict with attribute access; nested mappings become ConfigNodes.  Missing attributes raise
    AttributeError so `getattr(node, key, default)` / `hasattr` behave like on the reference dataclasses."""

    def __init__(self, data: Mapping | None = None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    def __setitem__(self, key, value):
        super().__setitem__(key, ConfigNode(value) if isinstance(value, Mapping) and not isinstance(value, ConfigNode)
                            else value)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        self[key] = value

    def __deepcopy__(self, memo):
        return ConfigNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self) -> dict:
        return {k: (v.to_dict() if isinstance(v, ConfigNode) else copy.deepcopy(v)) for k, v in self.items()}


Config = ConfigNode


def _deep_merge(dst: dict, src: Mapping) -> dict:
    for k, v in src.items():
        if isinstance(v, Mapping) and isinstance(dst.get(k), Mapping):
            _deep_merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v) if not isinstance(v, Mapping) else _deep_merge({}, v)
    return dst


def _load_with_bases(path: Path, seen: tuple = ()) -> dict:
    path = path.resolve()
    if path in seen:
        raise ValueError(f"circular _base_ inheritance through {path}")
    with open(path) as fh:
        raw = yaml.safe_load(fh) or {}
    if not isinstance(raw, Mapping):
        raise ValueError(f"{path}: top level of a config must be a mapping")
    merged: dict = {}
    bases = raw.get("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    for b in bases or []:
        bp = (path.parent / b)
        if not bp.exists():
            warnings.warn(f"config base {b!r} (from {path.name}) not found; skipped (profile libraries of the "
                          "reference are not shipped with this engine)")
            continue
        _deep_merge(merged, _load_with_bases(bp, seen + (path,)))
    _deep_merge(merged, {k: v for k, v in raw.items() if k != "_base_"})
    return merged


def sync_inference_runtime_aliases(cfg: ConfigNode, user_window: Mapping | None = None) -> None:
    """Copy the canonical `inference.window` values the user set into `inference.sliding_window`."""
    inf = cfg.get("inference")
    if inf is None:
        return
    win, sw = inf.get("window"), inf.get("sliding_window")
    if win is not None and sw is not None:
        keys = _WINDOW_KEYS if user_window is None else [k for k in _WINDOW_KEYS if k in user_window]
        for k in keys:
            sw[k] = copy.deepcopy(win[k])
    model = inf.get("model")
    if model is not None:
        for k in ("head", "select_channel", "crop_pad"):
            if model.get(k) is not None:
                inf[k] = model[k]
    ex = inf.get("execution")
    if ex is not None and ex.get("strategy") is not None:
        inf["strategy"] = ex["strategy"]


def resolve_default_profiles(raw: Mapping, mode: str = "train") -> dict:
    """runtime tree = top-level sections  <-  `default`  <-  the `mode` stage section."""
    stage_key = {"tune-test": "test"}.get(mode, mode)
    out: dict = {}
    _deep_merge(out, {k: v for k, v in raw.items() if k not in ("default", "train", "test", "tune", "tune_test")})
    for section in ("default", stage_key):
        sec = raw.get(section)
        if isinstance(sec, Mapping):
            _deep_merge(out, sec)
    return out


def load_config(path: str | Path, mode: str = "train", overrides: Iterable[str] = ()) -> ConfigNode:
    raw = _load_with_bases(Path(path))
    # profile libraries pulled in through _base_ (e.g. all_profiles.yaml) are tolerated but not expanded
    for k in [k for k in raw if k.endswith("_profiles") or k.endswith("_templates")]:
        raw.pop(k)
    unknown = sorted(set(raw) - _TOP_LEVEL)
    if unknown:
        raise ValueError(f"Unknown top-level config keys {unknown} in {path}. Allowed: {sorted(_TOP_LEVEL)}")
    user = resolve_default_profiles(raw, mode)
    tree = schema_defaults()
    _deep_merge(tree, user)
    cfg = ConfigNode(tree)
    update_from_cli(cfg, overrides)
    inf_user = user.get("inference", {}) if isinstance(user.get("inference"), Mapping) else {}
    sync_inference_runtime_aliases(cfg, inf_user.get("window") if isinstance(inf_user.get("window"), Mapping) else {})
    for ov in overrides:   # explicit CLI writes to inference.sliding_window.* win over the alias sync
        if ov.startswith("inference.sliding_window."):
            update_from_cli(cfg, [ov])
    validate_config(cfg)
    return cfg


def update_from_cli(cfg: ConfigNode, overrides: Iterable[str]) -> ConfigNode:
    for item in overrides:
        if "=" not in item:
            raise ValueError(f"override {item!r} must look like key.sub=value")
        key, text = item.split("=", 1)
        value = yaml.safe_load(text)
        node = cfg
        parts = key.strip().split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], Mapping):
                node[p] = ConfigNode()
            node = node[p]
        node[parts[-1]] = value
    return cfg


def validate_config(cfg: ConfigNode) -> None:
    arch = cfg.model.arch.type
    if not isinstance(arch, str) or not arch:
        raise ValueError("model.arch.type must be a non-empty string")
    ov = cfg.inference.sliding_window.overlap
    if isinstance(ov, (int, float)) and not 0 <= float(ov) < 1:
        raise ValueError(f"inference.window.overlap must be in [0, 1), got {ov}")
    ws = cfg.inference.sliding_window.window_size
    if ws is not None and (len(ws) not in (2, 3) or any(int(v) <= 0 for v in ws)):
        raise ValueError(f"inference.window.window_size must be 2 or 3 positive ints, got {ws}")


__all__ = ["Config", "ConfigNode", "load_config", "resolve_default_profiles", "sync_inference_runtime_aliases",
           "update_from_cli", "validate_config", "schema_defaults"]
