"""Training step of the hot path on HIP kernels (forward + backward of the MedNeXt trunk as autograd Functions)
and the Lightning-free ConnectomicsModule / fit loop."""
from .autograd import mednext_train_forward
from .module import ConnectomicsModule, WarmupCosineLR, build_optimizer, fit, load_ema_state_dict, synthetic_batches

__all__ = ["mednext_train_forward", "ConnectomicsModule", "WarmupCosineLR", "build_optimizer", "fit",
           "load_ema_state_dict", "synthetic_batches"]
