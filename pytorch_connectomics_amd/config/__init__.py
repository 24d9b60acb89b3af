"""Config surface of the hot path -- a small PyYAML loader for the keys SURVEY.md section 5.6 lists
(the reference uses OmegaConf dataclasses, config/pipeline/config_io.py:264-297; OmegaConf is not in the image).

Supported: `_base_` inheritance (relative paths, missing bases are skipped with a warning), the reference's YAML
PROFILE ENGINE (config/pipeline/profile_engine.py: selectors such as `model.arch.profile: rsunet` expand the
registries -- `arch_profiles`, `loss_profiles`, `optimizer_profiles`, `activation_profiles`, ... -- that a config pulls in
through `_base_: ../connectomics/config/all_profiles.yaml`; the registries themselves are the reference's data files and
are read from wherever the user's `_base_` points, they are not shipped here), the
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
        "system": {"profile": None, "num_gpus": 1, "num_workers": 8, "seed": 42, "accelerator": "auto"},      # schema/system.py (8 readers)
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
                     "deep_supervision_clamp_min": -20.0, "deep_supervision_clamp_max": 20.0, "losses": None,
                     # schema/model.py:13-19 (LossBalancingConfig): None | "uncertainty" | "gradnorm" (training/balancing.py)
                     "loss_balancing": {"strategy": None, "gradnorm_alpha": 0.5, "gradnorm_lambda": 1.0,
                                        "gradnorm_parameter_strategy": "last"}},
        },
        "data": {"train": {"image": None, "label": None, "do_2d": False},
                 "val": {"image": None, "label": None, "do_2d": False},
                 "test": {"image": None, "label": None, "mask": None},
                 # schema/data.py: batch_size 4 (also the sliding-window batch when sw_batch_size is unset, window.py:413-423) and image
                 # normalisation "0-1" are the reference's defaults -- tutorials such as mito_lucchi++ rely on them unnamed.  NOT taken
                 # over: dataloader.patch_size / model.input_size / model.output_size = [128, 128, 128] (here None: a configuration
                 # names its patch and window sizes) and model.arch.type = monai_basic_unet3d (outside the hot path; here mednext)
                 "dataloader": {"batch_size": 4, "patch_size": None, "use_lazy_zarr": False, "use_lazy_h5": False},
                 "data_transform": {"patch_size": None},
                 "image_transform": {"transform_profile": None, "normalize": "0-1", "clip_percentile_low": 0.0, "clip_percentile_high": 1.0},
                 "mask_transform": None,
                 # schema/data.py:66-86 (the keys the inference path reads: stacked label targets -> affinity channel groups)
                 "label_transform": {"keys": ["label"], "stack_outputs": True, "retain_original": False, "output_dtype": "float32",
                                     "targets": []}},
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
                         "save_intermediate": False, "precomputed": False, "precomputed_resolution": None,
                         "precomputed_chunk_size": [128, 128, 64], "precomputed_affinity_convention": "none"},
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


class Config(ConfigNode):
    """`Config()` = the schema defaults of the sections the engine reads, like the reference's dataclass constructor
    (`cfg = Config(); cfg.inference.sliding_window.window_size = [...]`); `Config(mapping)` wraps the mapping as it is."""

    def __init__(self, data: Mapping | None = None):
        super().__init__(schema_defaults() if data is None else data)


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


# ---------------------------------------------------------------------------------------------- profile engine
# (profiles_key, stages, selector path relative to the stage, target path relative to the stage, list key) -- the table of
# config/pipeline/profile_engine.py:484-551; "" = the stage section itself.  Order matters: pipeline / system, arch, the rest.
_STAGES = ("default", "train", "test", "tune")
_VALUE_PROFILE_FAMILIES = [
    ("pipeline_profiles", ("default",), "pipeline_profile", "", None),
    ("system_profiles", _STAGES, "system.profile", "system", None),
    ("arch_profiles", _STAGES, "model.arch.profile", "model", None),
    ("augmentation_profiles", ("default", "train"), "data.augmentation.profile", "data.augmentation", None),
    ("dataloader_profiles", _STAGES, "data.dataloader.profile", "data.dataloader", None),
    ("optimizer_profiles", ("default", "train"), "optimization.profile", "optimization", None),
    ("loss_profiles", ("default", "train"), "model.loss.profile", "model.loss", "losses"),
    ("label_profiles", ("default", "train"), "data.label_transform.profile", "data.label_transform", None),
    ("activation_profiles", ("default", "test", "tune"), "inference.model.activation_profile", "inference.model",
     "channel_activations"),
    ("tune_profiles", ("tune",), "profile", "", None),
]
# `${profiles_key.name}` string references, resolved at these stage-relative paths (profile_engine.py:578-587)
_REFERENCE_PROFILE_FAMILIES = [("loss_profiles", ("default", "train"), "model.loss"),
                               ("label_profiles", ("default", "train"), "data.label_transform"),
                               ("activation_profiles", ("default", "test", "tune"), "inference.model"),
                               ("augmentation_profiles", ("default", "train"), "data.augmentation")]


def _join(stage: str, rel: str) -> str:
    return f"{stage}.{rel}" if (stage and rel) else (stage or rel)


def _select(tree: Mapping, path: str):
    node: Any = tree
    for part in [p for p in path.split(".") if p]:
        if not isinstance(node, Mapping) or part not in node:
            return None
        node = node[part]
    return node


def _assign(tree: dict, path: str, value) -> None:
    parts = [p for p in path.split(".") if p]
    if not parts:                                    # the stage root itself: merge the mapping in place
        tree.clear()
        tree.update(value)
        return
    node = tree
    for part in parts[:-1]:
        if not isinstance(node.get(part), dict):
            node[part] = {}
        node = node[part]
    node[parts[-1]] = value


def _allowed_selector_paths() -> set:
    return {_join(stage, sel) for _k, stages, sel, _t, _l in _VALUE_PROFILE_FAMILIES for stage in stages}


def _reject_noncanonical_selectors(raw: Mapping) -> None:
    """A `profile` / `*_profile` key with a value anywhere but at a selector path of the table is an error
    (profile_engine.py:150-164): a typo there would otherwise silently train the default architecture."""
    allowed = _allowed_selector_paths()
    parents = {sel.rsplit(".", 1)[0] if "." in sel else "" for _k, _s, sel, _t, _l in _VALUE_PROFILE_FAMILIES if sel.endswith("profile")}
    found = set()

    def walk(node, path, in_registry):
        if isinstance(node, list):
            for item in node:
                walk(item, path, in_registry)
            return
        if not isinstance(node, Mapping):
            return
        for key, child in node.items():
            key = str(key)
            child_path = f"{path}.{key}" if path else key
            reg = in_registry or (path == "" and (key.endswith("_profiles") or key.endswith("_templates")))
            if not reg and child not in (None, ""):
                if key.endswith("_profile"):
                    found.add(child_path)
                elif key == "profile":
                    parent = path
                    head, _, tail = parent.partition(".")
                    rel = tail if head in _STAGES else parent
                    if head in _STAGES and not tail:
                        rel = ""
                    if rel in parents:
                        found.add(child_path)
            walk(child, child_path, reg)

    walk(raw, "", False)
    bad = sorted(p for p in found if p not in allowed)
    if bad:
        raise ValueError("Non-canonical profile selector path(s) detected: " + ", ".join(bad) +
                         ". Allowed selector paths: [" + ", ".join(sorted(allowed)) + "]")


def apply_profiles(raw: dict) -> dict:
    """Expand profile selectors in place (profile_engine.py ValueProfileApplier / ReferenceProfileApplier /
    YamlProfileEngine): for every family and stage, `stage.<selector>: name` merges `<profiles_key>[name]` into
    `stage.<target>` with the values already written there WINNING over the profile payload (lists replace, mappings merge);
    an `overrides: {index: patch}` mapping next to the selector patches entries of the expanded list; `${key.name}` strings at
    the reference paths are replaced by the payload; the registries are removed afterwards."""
    if "shared" in raw:
        raise ValueError("Top-level 'shared' config section has been removed. Use top-level 'default' instead.")
    _reject_noncanonical_selectors(raw)
    for key, stages, sel_rel, tgt_rel, list_key in _VALUE_PROFILE_FAMILIES:
        for stage in stages:
            sel_path, tgt_path = _join(stage, sel_rel), _join(stage, tgt_rel)
            selected = _select(raw, sel_path)
            if selected is None:
                continue
            profiles = raw.get(key)
            if profiles is None:
                raise ValueError(f"Selector '{selected}' at '{sel_path}' requires '{key}' to be defined in YAML.")
            if selected not in profiles:
                raise ValueError(f"Unknown selector '{selected}' at '{sel_path}'. Available profiles: ["
                                 + ", ".join(sorted(str(k) for k in profiles)) + "]")
            payload = copy.deepcopy(profiles[selected])
            existing = _select(raw, tgt_path)
            if isinstance(existing, Mapping) and isinstance(payload, Mapping):
                merged = _deep_merge(payload, copy.deepcopy(dict(existing)))      # explicit values win
            else:
                merged = payload if existing is None else existing
            _assign(raw, tgt_path, merged)
            # positional overrides of a list-valued profile (profile_engine.py:207-247)
            parent = sel_path.rsplit(".", 1)[0] if "." in sel_path else ""
            ov = _select(raw, _join(parent, "overrides"))
            if isinstance(ov, Mapping):
                lst = _select(raw, _join(tgt_path, list_key) if list_key else tgt_path)
                if isinstance(lst, list):
                    for idx_key, patch in ov.items():
                        idx = int(idx_key)
                        if idx < 0 or idx >= len(lst):
                            raise ValueError(f"Override index {idx} at '{_join(parent, 'overrides')}' is out of range for profile "
                                             f"list at '{tgt_path}' (length {len(lst)}).")
                        if isinstance(patch, Mapping) and isinstance(lst[idx], Mapping):
                            lst[idx] = _deep_merge(copy.deepcopy(dict(lst[idx])), dict(patch))
                    holder = _select(raw, parent) if parent else raw
                    if isinstance(holder, dict):
                        holder.pop("overrides", None)
    import re
    for key, stages, tgt_rel in _REFERENCE_PROFILE_FAMILIES:
        profiles = raw.get(key)
        if profiles is None:
            continue
        pat = re.compile(r"\$\{" + re.escape(key) + r"\.([A-Za-z0-9_\-]+)\}")
        for stage in stages:
            value = _select(raw, _join(stage, tgt_rel))
            m = pat.fullmatch(value) if isinstance(value, str) else None
            if m:
                if m.group(1) not in profiles:
                    raise ValueError(f"Unknown profile '{m.group(1)}' in {key}. Available profiles: ["
                                     + ", ".join(sorted(str(k) for k in profiles)) + "]")
                _assign(raw, _join(stage, tgt_rel), copy.deepcopy(profiles[m.group(1)]))
    for k in [k for k in raw if k.endswith("_profiles") or k.endswith("_templates")]:
        raw.pop(k)
    return raw


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
    apply_profiles(raw)          # selectors -> payloads of the registries pulled in through _base_; registries removed
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


__all__ = ["Config", "ConfigNode", "load_config", "apply_profiles", "resolve_default_profiles", "sync_inference_runtime_aliases",
           "update_from_cli", "validate_config", "schema_defaults"]
