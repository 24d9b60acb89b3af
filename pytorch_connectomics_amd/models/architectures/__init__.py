"""Architecture registry + the hot-path model builders (MedNeXt; RSUNet; MONAI-style residual U-Net)."""
from .base import ConnectomicsModel
from .registry import (get_architecture_builder, get_architecture_info, is_architecture_available,
                       list_architectures, register_architecture, unregister_architecture)
from . import mednext_models  # noqa: F401  (registers 'mednext', 'mednext_custom')
from . import rsunet  # noqa: F401          (registers 'rsunet', 'rsunet_iso')
from . import monai_models  # noqa: F401    (registers 'monai_unet')
from .mednext_models import MedNeXtMultiHeadWrapper, MedNeXtTaskHead, MedNeXtWrapper


def get_available_architectures() -> dict:
    """Registered architectures by family, the reference's dictionary (architectures/__init__.py:70-90): every family of this
    package is built in (no optional third-party dependency decides what is registered); `nnunet` has no builder here."""
    names = list_architectures()
    families = {family: [a for a in names if a.startswith(prefix)]
                for family, prefix in (("monai", "monai_"), ("mednext", "mednext"), ("rsunet", "rsunet"), ("nnunet", "nnunet"))}
    return {"all": names, **families}


def print_available_architectures() -> None:
    for name in list_architectures():
        print(name)


__all__ = ["ConnectomicsModel", "register_architecture", "get_architecture_builder", "list_architectures",
           "is_architecture_available", "unregister_architecture", "get_architecture_info",
           "get_available_architectures", "print_available_architectures", "MedNeXtWrapper", "MedNeXtTaskHead", "MedNeXtMultiHeadWrapper"]
