"""Channel selector parsing: ints, "a:b" half-open slice strings (no step), or explicit int lists --
the contract of the reference's connectomics/utils/channel_slices.py (negative indices follow Python
rules; out-of-range / empty selections raise ValueError naming the selector)."""
from __future__ import annotations

from typing import Any, Optional, Sequence, Union

ChannelRangeSelector = Union[int, str]                    # one channel or one 'a:b' slice string
ChannelSelector = Union[int, str, Sequence[int]]          # ... or an explicit channel list


def _parse(text: str, context: str):
    t = text.strip()
    if not t:
        raise ValueError(f"{context} must not be empty.")
    if ":" not in t:
        try:
            return int(t)
        except ValueError as exc:
            raise ValueError(f"{context} must be an integer index or a Python-style slice string, "
                             f"got {text!r}.") from exc
    if t.count(":") != 1:
        raise ValueError(f"{context} must use step-free Python slice syntax 'start:end', got {text!r}.")
    a, b = (p.strip() for p in t.split(":", 1))
    try:
        return slice(int(a) if a else None, int(b) if b else None)
    except ValueError as exc:
        raise ValueError(f"{context} must use integer slice bounds in 'start:end', got {text!r}.") from exc


def normalize_channel_selector(selector: Any, *, context: str = "channel selector"):
    """None | int | 'a:b' string | list[int]  (strings holding a single int become ints)."""
    if selector is None:
        return None
    if isinstance(selector, int):            # bool is an int here, as in the reference (True selects channel 1)
        return int(selector)
    if isinstance(selector, str):
        parsed = _parse(selector, context)
        if isinstance(parsed, int):
            return parsed
        return f"{'' if parsed.start is None else parsed.start}:{'' if parsed.stop is None else parsed.stop}"
    if isinstance(selector, Sequence) and not isinstance(selector, (str, bytes)):
        if len(selector) == 0:
            raise ValueError(f"{context} must not be an empty channel list.")
        out = []
        for raw in selector:
            if isinstance(raw, int):
                out.append(int(raw))
            elif isinstance(raw, str):
                try:
                    out.append(int(raw.strip()))
                except ValueError as exc:
                    raise ValueError(f"{context} channel lists must contain only integer indices, "
                                     f"got {raw!r}.") from exc
            else:
                raise TypeError(f"{context} channel lists must contain only integers, got {type(raw).__name__}.")
        return out
    raise TypeError(f"{context} must be an int, a Python-style slice string, or an explicit list of ints; "
                    f"got {type(selector).__name__}.")


def resolve_channel_index(index_value: int, *, num_channels: int, context: str = "channel selector") -> int:
    idx = int(index_value)
    if idx < 0:
        idx += num_channels
    if not 0 <= idx < num_channels:
        raise ValueError(f"Invalid {context} {index_value!r} for tensor with {num_channels} channels: "
                         f"resolved index {idx} is out of bounds.")
    return idx


def resolve_channel_range(selector, *, num_channels: int, context: str = "channel selector") -> tuple[int, int]:
    """Contiguous selector -> absolute half-open (start, stop)."""
    if num_channels <= 0:
        raise ValueError(f"{context} requires num_channels > 0, got {num_channels}.")
    if selector is not None and not isinstance(selector, (int, str)):       # a range is one int or one slice string
        raise TypeError(f"{context} must be an int or a Python-style slice string, got {type(selector).__name__}.")
    norm = normalize_channel_selector(selector, context=context)
    if norm is None:
        return 0, num_channels
    if isinstance(norm, int):
        i = resolve_channel_index(norm, num_channels=num_channels, context=context)
        return i, i + 1
    sl = _parse(norm, context)
    start = 0 if sl.start is None else int(sl.start)
    stop = num_channels if sl.stop is None else int(sl.stop)
    if start < 0:
        start += num_channels
    if stop < 0:
        stop += num_channels
    if not 0 <= start < num_channels:
        raise ValueError(f"Invalid {context} {norm!r} for tensor with {num_channels} channels: "
                         f"resolved start index {start} is out of bounds.")
    if not 0 <= stop <= num_channels:
        raise ValueError(f"Invalid {context} {norm!r} for tensor with {num_channels} channels: "
                         f"resolved stop index {stop} is out of bounds.")
    if stop <= start:
        raise ValueError(f"Invalid {context} {norm!r} for tensor with {num_channels} channels: "
                         f"resolved range [{start}, {stop}) is empty or inverted.")
    return start, stop


def resolve_channel_indices(selector, *, num_channels: int, context: str = "channel selector") -> list[int]:
    if num_channels <= 0:
        raise ValueError(f"{context} requires num_channels > 0, got {num_channels}.")
    norm = normalize_channel_selector(selector, context=context)
    if norm is None:
        return list(range(num_channels))
    if isinstance(norm, list):
        return [resolve_channel_index(i, num_channels=num_channels, context=context) for i in norm]
    if isinstance(norm, int):
        return [resolve_channel_index(norm, num_channels=num_channels, context=context)]
    a, b = resolve_channel_range(norm, num_channels=num_channels, context=context)
    return list(range(a, b))


def normalize_channel_range_selector(selector: Any, *, context: str = "channel selector") -> Optional[ChannelRangeSelector]:
    """A CONTIGUOUS selector in canonical form: None (all channels), an int, or an 'a:b' string; channel lists are refused."""
    if selector is None or isinstance(selector, (int, str)):
        return normalize_channel_selector(selector, context=context)
    raise TypeError(f"{context} must be an int or a Python-style slice string, got {type(selector).__name__}.")


def infer_min_required_channels(selector: Any, *, context: str = "channel selector") -> Optional[int]:
    """The smallest channel count for which `selector` is valid (None selects everything: no requirement).  An index i needs
    i + 1 channels (-i needs i); a list needs what its most demanding entry needs; a slice is tried against growing channel
    counts up to |bound| + 2, the first count it resolves against without an empty selection wins."""
    norm = normalize_channel_selector(selector, context=context)
    if norm is None:
        return None
    if isinstance(norm, list):
        return max(infer_min_required_channels(entry, context=context) or 1 for entry in norm)
    if isinstance(norm, int):
        return norm + 1 if norm >= 0 else -norm
    piece = _parse(norm, context)
    reach = max([1] + [abs(bound) + 2 for bound in (piece.start, piece.stop) if bound is not None])
    for count in range(1, reach + 1):
        try:
            resolve_channel_range(norm, num_channels=count, context=context)
        except ValueError:
            continue
        return count
    raise ValueError(f"Could not infer a valid channel count for {context} {norm!r}.")


__all__ = ["ChannelRangeSelector", "ChannelSelector", "normalize_channel_selector", "normalize_channel_range_selector",
           "resolve_channel_index", "resolve_channel_range", "resolve_channel_indices", "infer_min_required_channels"]
