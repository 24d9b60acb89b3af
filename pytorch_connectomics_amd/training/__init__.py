"""Training step of the hot path on HIP kernels (forward + backward of the MedNeXt trunk as autograd Functions)."""
from .autograd import mednext_train_forward

__all__ = ["mednext_train_forward"]
