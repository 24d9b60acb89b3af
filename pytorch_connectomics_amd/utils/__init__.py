"""connectomics.utils counterpart: channel selectors and model-output selection."""
from .channel_slices import (normalize_channel_selector, resolve_channel_index, resolve_channel_indices,
                             resolve_channel_range)
from .model_outputs import resolve_output_head, resolve_output_heads, select_output_tensor, unwrap_main_output

__all__ = ["normalize_channel_selector", "resolve_channel_index", "resolve_channel_indices",
           "resolve_channel_range", "select_output_tensor", "unwrap_main_output", "resolve_output_head", "resolve_output_heads"]
