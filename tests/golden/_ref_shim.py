"""Import shim that loads the *reference's own* hot-path modules in this container.

Only used by tests/golden/make_golden.py (fixture generation, run where
/root/reference exists).  Nothing here travels to the GPU box at run time: the
fixtures it produces are data (inputs + expected outputs).

The reference package cannot be imported normally here (monai / lightning /
omegaconf / h5py are absent), so we register empty stub *packages* with
``__path__`` pointing into /root/reference and then import the handful of
torch-only modules that hold the real arithmetic.
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

REF_ROOT = Path("/root/reference")


def _stub_pkg(name: str, rel: str | None = None) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__path__ = [str(REF_ROOT / (rel if rel is not None else name.replace(".", "/")))]
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


def install() -> None:
    if "connectomics" in sys.modules and getattr(sys.modules["connectomics"], "_pytc_shim", False):
        return
    if not REF_ROOT.exists():
        raise RuntimeError("/root/reference is not available; fixtures can only be regenerated "
                           "in the build container")
    root = _stub_pkg("connectomics")
    root._pytc_shim = True
    for name in (
        "connectomics.models",
        "connectomics.models.architectures",
        "connectomics.models.losses",          # stub package: only losses.py (torch-only) is imported, not build.py (MONAI)
        "connectomics.inference",
        "connectomics.config",
        "connectomics.data",
        "connectomics.data.processing",
        "connectomics.data.augmentation",
        "connectomics.data.io",
        "connectomics.utils",
        "connectomics.training",
        "connectomics.training.optimization",
    ):
        _stub_pkg(name)
    # connectomics.config.hardware: only resolve_accelerator_type / empty cache are read
    hw = types.ModuleType("connectomics.config.hardware")
    hw.resolve_accelerator_type = lambda requested="auto": "cpu"
    hw.empty_accelerator_cache = lambda *a, **k: None
    sys.modules["connectomics.config.hardware"] = hw
    sys.modules["connectomics.config"].hardware = hw
    sys.modules["connectomics.config"].Config = type("Config", (), {})
    # omegaconf dummy for manager.py
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        oc.DictConfig = type("DictConfig", (dict,), {})
        oc.ListConfig = type("ListConfig", (list,), {})
        oc.OmegaConf = type("OmegaConf", (), {"to_container": staticmethod(lambda v, resolve=True: v)})
        sys.modules["omegaconf"] = oc
    # lazy.py leaf imports
    aug = types.ModuleType("connectomics.data.augmentation.augment_ops")
    aug.smart_normalize = lambda x, *a, **k: x
    sys.modules["connectomics.data.augmentation.augment_ops"] = aug
    io_mod = types.ModuleType("connectomics.data.io.io")
    io_mod._detect_format = lambda *a, **k: "numpy"
    io_mod._get_tiff_volume_shape = lambda *a, **k: None
    io_mod._tiff_series_are_stackable = lambda *a, **k: False
    sys.modules["connectomics.data.io.io"] = io_mod
    # inference/output.py: only the numpy dtype / intensity transforms are used; its MONAI-based resampler is not
    npp = types.ModuleType("connectomics.data.processing.nnunet_preprocess")
    npp.restore_prediction_to_input_space = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not available in the shim"))
    sys.modules["connectomics.data.processing.nnunet_preprocess"] = npp


def ref(name: str):
    install()
    return importlib.import_module(name)
