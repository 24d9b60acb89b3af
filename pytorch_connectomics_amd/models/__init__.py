"""connectomics.models counterpart: build_model + architecture registry."""
from .architectures import (ConnectomicsModel, get_architecture_builder, get_architecture_info,
                            is_architecture_available, list_architectures, print_available_architectures,
                            register_architecture, unregister_architecture)
from .build import build_model

__all__ = ["build_model", "ConnectomicsModel", "register_architecture", "get_architecture_builder",
           "list_architectures", "is_architecture_available", "unregister_architecture",
           "get_architecture_info", "print_available_architectures"]
