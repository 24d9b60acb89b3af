"""Architecture registry + the hot-path model builders (MedNeXt; RSUNet; MONAI-style residual U-Net)."""
from .base import ConnectomicsModel
from .registry import (get_architecture_builder, get_architecture_info, is_architecture_available,
                       list_architectures, register_architecture, unregister_architecture)
from . import mednext_models  # noqa: F401  (registers 'mednext', 'mednext_custom')
from . import rsunet  # noqa: F401          (registers 'rsunet', 'rsunet_iso')
from . import monai_models  # noqa: F401    (registers 'monai_unet')
from .mednext_models import MedNeXtMultiHeadWrapper, MedNeXtTaskHead, MedNeXtWrapper


def print_available_architectures() -> None:
    for name in list_architectures():
        print(name)


__all__ = ["ConnectomicsModel", "register_architecture", "get_architecture_builder", "list_architectures",
           "is_architecture_available", "unregister_architecture", "get_architecture_info",
           "print_available_architectures", "MedNeXtWrapper", "MedNeXtTaskHead", "MedNeXtMultiHeadWrapper"]
