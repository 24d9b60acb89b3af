"""Immediate-mode Python wrappers over the C ABI (torch tensors in, raw pointers out).

PyTorch is used here only as the owner of device memory and of the HIP stream; every
arithmetic op below is a hand-written gfx950 kernel in libpytc_hip.so.  All functions raise
if handed CPU tensors: there is no CPU path in the product.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _native as nat

_DT = {torch.float32: nat.F32, torch.bfloat16: nat.BF16}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"pytorch_connectomics_amd kernels support float32/bfloat16, got {dt}") from None


def require_device(device, what: str = "this inference path") -> None:
    """The engines call this before touching data: anything but a HIP device is an error (there is no CPU path)."""
    if torch.device(device).type != "cuda":
        raise RuntimeError(f"{what} (pytorch_connectomics_amd) needs a CUDA(HIP) device: there is no CPU path")


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(HIP) tensor: pytorch_connectomics_amd has no CPU path")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Stream object per call
    (8 us: a fifth of the host time of a launch-bound pass, profiles/r06_c3_tta16_host.txt); the raw getter is what it wraps."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg) -------
class _Profiler:
    def __init__(self):
        self.enabled = False
        self.records = []   # (name, start_event, end_event, algorithmic_bytes, flops, kernel symbol)

    def summary(self):
        """label -> launches / ms / algorithmic bytes / flops / the device kernel family the label dispatches to."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, nbytes, flops, symbol in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0, "symbol": symbol})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["bytes"] += nbytes
            d["flops"] += flops
        return out

    def by_symbol(self):
        """The same records grouped by device kernel family (one rocprof symbol, or one template): a kernel that runs under
        several per-shape labels shows up with its whole share of the step."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, nbytes, flops, symbol in self.records:        # per RECORD: launches of one label may be different template
            d = out.setdefault(symbol, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0, "labels": []})      # instances (mixer +head ...)
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["bytes"] += nbytes
            d["flops"] += flops
            if name not in d["labels"]:
                d["labels"].append(name)
        return out


PROFILER = _Profiler()


class profiled:
    """with profiled(): ... -> PROFILER.summary() gives per-kernel launch counts / ms / algorithmic bytes."""

    def __enter__(self):
        PROFILER.records = []
        PROFILER.enabled = True
        return PROFILER

    def __exit__(self, *exc):
        PROFILER.enabled = False
        return False


def _run(name: str, nbytes: int, fn, *args, flops: int = 0, symbol: Optional[str] = None) -> None:
    if PROFILER.enabled:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        st = fn(*args)
        e.record()
        PROFILER.records.append((name, s, e, int(nbytes), int(flops), symbol or name.split("[")[0]))
    else:
        st = fn(*args)
    nat.check(st, name)


def _nbytes(*ts) -> int:
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def _starts_array(starts: Sequence[Sequence[int]], windows: Optional[int] = None):
    """Window positions as a flat int32 array; with `windows`, refuses a batch whose size is not the number of positions (a network
    that dropped or added samples would otherwise blend the wrong windows, or read past the array)."""
    if windows is not None and len(starts) != int(windows):
        raise ValueError(f"{int(windows)} window predictions for {len(starts)} window positions: the network must keep the batch size")
    flat = [int(v) for s in starts for v in s]
    return (C.c_int32 * len(flat))(*flat)


# ------------------------------------------------------------------ sliding window
def gather_windows(vol: torch.Tensor, starts, roi, *, view: int = 0, pad_mode: str = "constant",
                   cval: float = 0.0, out_dtype: torch.dtype = torch.float32,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """vol fp32 (C,Z,Y,X) -> (B, rz, ry, rx, C) window batch (NDHWC)."""
    _dev(vol, "vol")
    if vol.dtype != torch.float32 or vol.dim() != 4:
        raise ValueError("vol must be float32 with shape (C, Z, Y, X)")
    Cc, Z, Y, X = vol.shape
    B = len(starts)
    rz, ry, rx = (int(v) for v in roi)
    if out is None:
        out = torch.empty((B, rz, ry, rx, Cc), dtype=out_dtype, device=vol.device)
    st = _starts_array(starts)
    for b0 in range(0, B, 64):
        nb = min(64, B - b0)
        sub = (C.c_int32 * (3 * nb)).from_buffer(st, 4 * 3 * b0)
        o = out[b0:b0 + nb]
        _run("gather_windows", 2 * _nbytes(o) if o.dtype == torch.float32 else 3 * _nbytes(o),
             nat.lib().pytc_gather_windows, _p(vol), Cc, Z, Y, X, sub, nb, rz, ry, rx, int(view),
             nat.PAD_MODES[pad_mode], float(cval), _p(o), dtype_code(out.dtype), _stream())
    return out


def blend_accumulate(pred: torch.Tensor, starts, value: torch.Tensor, weight: Optional[torch.Tensor],
                     wz: torch.Tensor, wy: torch.Tensor, wx: torch.Tensor, *, view: int = 0,
                     combine: int = nat.BLEND_PRODUCT, floor_w: float = 1e-5, border=None) -> None:
    """pred (B, rz, ry, rx, C) NDHWC; value (C,Z,Y,X) += pred*w; weight (Z,Y,X) += w."""
    _dev(pred, "pred"); _dev(value, "value")
    B, rz, ry, rx, Cc = pred.shape
    if value.dtype != torch.float32 or value.shape[0] != Cc:
        raise ValueError("value accumulator must be float32 (C, Z, Y, X) with C matching pred")
    _, Z, Y, X = value.shape
    st = _starts_array(starts, B)
    bd = (C.c_int32 * 3)(*[int(v) for v in border]) if border else None
    win = B * rz * ry * rx
    _run("blend_accumulate", _nbytes(pred) + win * 4 * (2 * Cc + (2 if weight is not None else 0)),
         nat.lib().pytc_blend_accumulate, _p(pred), dtype_code(pred.dtype), B, st, rz, ry, rx, Cc, int(view),
         _p(wz), _p(wy), _p(wx), int(combine), float(floor_w), bd, _p(value), _p(weight), Z, Y, X, _stream())


def blend_accumulate_mapped(pred: torch.Tensor, starts, value: torch.Tensor, weight: Optional[torch.Tensor],
                            wz: torch.Tensor, wy: torch.Tensor, wx: torch.Tensor, chan_src, chan_shift, *, view: int = 0,
                            combine: int = nat.BLEND_PRODUCT, floor_w: float = 1e-5, border=None) -> None:
    """blend_accumulate with the affinity channel map: output channel d <- pred channel chan_src[d] displaced by
    chan_shift[d] (window-local (z,y,x)); see pytc_blend_accumulate_mapped."""
    _dev(pred, "pred"); _dev(value, "value")
    B, rz, ry, rx, Cc = pred.shape
    if value.dtype != torch.float32 or value.shape[0] != Cc:
        raise ValueError("value accumulator must be float32 (C, Z, Y, X) with C matching pred")
    if len(chan_src) != Cc or len(chan_shift) != Cc:
        raise ValueError("channel map must describe every output channel")
    _, Z, Y, X = value.shape
    st = _starts_array(starts, B)
    bd = (C.c_int32 * 3)(*[int(v) for v in border]) if border else None
    cs = (C.c_int32 * Cc)(*[int(v) for v in chan_src])
    sh = (C.c_int32 * (3 * Cc))(*[int(v) for s3 in chan_shift for v in s3])
    win = B * rz * ry * rx
    _run("blend_accumulate_mapped", _nbytes(pred) + win * 4 * (2 * Cc + (2 if weight is not None else 0)),
         nat.lib().pytc_blend_accumulate_mapped, _p(pred), dtype_code(pred.dtype), B, st, rz, ry, rx, Cc, int(view),
         _p(wz), _p(wy), _p(wx), int(combine), float(floor_w), bd, cs, sh, _p(value), _p(weight), Z, Y, X, _stream())


def blend_weight_shifted(starts, roi, weight: torch.Tensor, wz, wy, wx, shift, *, combine: int = nat.BLEND_PRODUCT,
                         floor_w: float = 1e-5, border=None) -> None:
    """weight (Z,Y,X) += blending map over the box a window displaced by `shift` covers (per window, in order)."""
    _dev(weight, "weight")
    Z, Y, X = weight.shape
    st = _starts_array(starts)
    B = len(starts)
    bd = (C.c_int32 * 3)(*[int(v) for v in border]) if border else None
    sh = (C.c_int32 * 3)(*[int(v) for v in shift])
    _run("blend_weight_shifted", B * int(roi[0]) * int(roi[1]) * int(roi[2]) * 8, nat.lib().pytc_blend_weight_shifted, B, st,
         int(roi[0]), int(roi[1]), int(roi[2]), _p(wz), _p(wy), _p(wx), int(combine), float(floor_w), bd, sh, _p(weight),
         Z, Y, X, _stream())


def normalize_covered(value: torch.Tensor, weight: torch.Tensor) -> None:
    """value <- value / weight where weight > 0, else 0 (in place; same shapes)."""
    _dev(value, "value"); _dev(weight, "weight")
    if value.numel() != weight.numel() or value.dtype != torch.float32 or weight.dtype != torch.float32:
        raise ValueError("normalize_covered expects float32 tensors of equal size")
    _run("normalize_covered", 2 * _nbytes(value) + _nbytes(weight), nat.lib().pytc_normalize_covered, _p(value), _p(weight),
         value.numel(), _stream())


def ensemble_update_masked(stat: torch.Tensor, count: torch.Tensor, x: torch.Tensor, cover: Optional[torch.Tensor],
                           mode: int) -> None:
    _dev(stat, "stat"); _dev(x, "x")
    _run("ensemble_update_masked", 3 * _nbytes(stat) + 2 * _nbytes(count), nat.lib().pytc_ensemble_update_masked, _p(stat),
         _p(count), _p(x), _p(cover), stat.numel(), int(mode), _stream())


def ensemble_finalize_masked(stat: torch.Tensor, count: torch.Tensor, out: torch.Tensor, mode: int) -> None:
    _dev(stat, "stat"); _dev(out, "out")
    _run("ensemble_finalize_masked", 3 * _nbytes(stat), nat.lib().pytc_ensemble_finalize_masked, _p(stat), _p(count), _p(out),
         stat.numel(), int(mode), _stream())


def blend_finalize(value: torch.Tensor, weight: torch.Tensor, *, clamp: float = 1e-4, act: int = nat.ACT_NONE) -> None:
    _dev(value, "value"); _dev(weight, "weight")
    Cc = value.shape[0]
    nvox = weight.numel()
    _run("blend_finalize", 2 * _nbytes(value) + _nbytes(weight), nat.lib().pytc_blend_finalize, _p(value),
         _p(weight), Cc, nvox, float(clamp), int(act), _stream())


def channel_activation(value: torch.Tensor, c0: int, c1: int, act: int, scale: float = 1.0, *,
                       channels_last: bool = False) -> None:
    """value fp32, (C, *spatial) or (*, C) when channels_last: channels [c0, c1) <- act(scale * v) in place."""
    _dev(value, "value")
    if value.dtype != torch.float32:
        raise TypeError("channel_activation works on float32 volumes")
    Cc = value.shape[-1] if channels_last else value.shape[0]
    nvox = value.numel() // Cc
    _run("channel_activation", 2 * nvox * 4 * (c1 - c0), nat.lib().pytc_channel_activation, _p(value), Cc, nvox,
         int(channels_last), int(c0), int(c1), int(act), float(scale), _stream())


def ensemble_update(acc: torch.Tensor, x: torch.Tensor, mode: int, count: int) -> None:
    _dev(acc, "acc"); _dev(x, "x")
    _run("ensemble_update", 2 * _nbytes(acc) + _nbytes(x), nat.lib().pytc_ensemble_update, _p(acc), _p(x),
         acc.numel(), int(mode), int(count), _stream())


# ------------------------------------------------------------------ disk-backed volumes (csrc/volume_kernels.hip)
def resample_region(raw_bytes: torch.Tensor, raw_dtype: str, strides_czyx, channels: int, tab_i0: torch.Tensor,
                    tab_i1: torch.Tensor, tab_f: torch.Tensor, dims_zyx) -> torch.Tensor:
    """raw_bytes: the storage bytes of a raw box on the device (uint8 view of any stored dtype `raw_dtype`); tables: int32 / fp32
    device vectors of nz + ny + nx entries (see pytc_resample_region) -> fp32 (C, nz, ny, nx)."""
    _dev(raw_bytes, "raw_bytes"); _dev(tab_i0, "tab_i0"); _dev(tab_i1, "tab_i1"); _dev(tab_f, "tab_f")
    if raw_dtype not in nat.RAW_DTYPES:
        raise TypeError(f"resample_region: stored dtype {raw_dtype} is not supported (one of {sorted(nat.RAW_DTYPES)})")
    nz, ny, nx = (int(v) for v in dims_zyx)
    n = nz + ny + nx
    if tab_i0.numel() != n or tab_i1.numel() != n or tab_f.numel() != n or tab_i0.dtype != torch.int32 or tab_f.dtype != torch.float32:
        raise ValueError("resample_region: tables must be int32 / int32 / float32 vectors of nz + ny + nx entries")
    out = torch.empty((int(channels), nz, ny, nx), dtype=torch.float32, device=raw_bytes.device)
    st = (C.c_int64 * 4)(*[int(v) for v in strides_czyx])
    _run("resample_region", raw_bytes.numel() + _nbytes(out), nat.lib().pytc_resample_region, _p(raw_bytes), nat.RAW_DTYPES[raw_dtype],
         st, int(channels), _p(tab_i0), _p(tab_i1), _p(tab_f), _i3((nz, ny, nx)), _p(out), _stream())
    return out


def window_normalize(x: torch.Tensor, *, mode: int = nat.NORM_NONE, binarize: bool = False, threshold: float = 0.0,
                     divide: float = 1.0, clip: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x fp32 (B, ...) contiguous, IN PLACE: per window (= leading index) binarise, clip to clip[b] = (lo, hi), then
    NORM_ZSCORE / NORM_MINMAX with that window's own statistics, or NORM_DIVIDE (pytc_window_normalize)."""
    _dev(x, "x")
    if x.dtype != torch.float32:
        raise TypeError("window_normalize works on float32 windows")
    B = int(x.shape[0])
    n = x.numel() // max(B, 1)
    if clip is not None:
        _dev(clip, "clip")
        if tuple(clip.shape) != (B, 2) or clip.dtype != torch.float32:
            raise ValueError("window_normalize: clip must be float32 (B, 2)")
    ws = None
    if mode in (nat.NORM_ZSCORE, nat.NORM_MINMAX):
        ws = torch.empty((int(nat.lib().pytc_window_normalize_ws_elems(B, n)),), dtype=torch.float64, device=x.device)
    _run("window_normalize", 2 * _nbytes(x) * (2 if ws is not None else 1), nat.lib().pytc_window_normalize, _p(x), B, n, int(mode),
         int(bool(binarize)), float(threshold), float(divide), _p(clip), _p(ws), _stream())
    return x


# ------------------------------------------------------------------ depthwise conv + norm statistics
def dwconv3d(x: torch.Tensor, w_taps: torch.Tensor, bias: Optional[torch.Tensor], *, K: int, stride: int = 1,
             stats: bool = True, transposed: bool = False, y: Optional[torch.Tensor] = None, store: bool = True,
             wide_range: bool = False):
    """x (N,D,H,W,C) -> y, stats(N,slots,2,C)|None.  w_taps fp32 (K^3, C).  store=False (K = 3: transposed, or bf16 stride 1 on
    the matrix-core kernel): statistics only, y is returned as None (the fused block / up-block kernels recompute it).  wide_range: x is a GRADIENT (values far below the
    f16 range): the bf16 z-march kernel keeps fp32 partial sums (pytc_dwconv3d_fwd_wide)."""
    _dev(x, "x"); _dev(w_taps, "w_taps")
    N, D, H, W, Cc = x.shape
    dt = dtype_code(x.dtype)
    if transposed:
        oshape = (N, 2 * D, 2 * H, 2 * W, Cc)
    else:
        p = K // 2
        oshape = (N, (D + 2 * p - K) // stride + 1, (H + 2 * p - K) // stride + 1, (W + 2 * p - K) // stride + 1, Cc)
    if not store:
        # statistics only: the K = 3 transposed kernels and the bf16 matrix-core stride-1 kernel (the C ABI rejects other shapes)
        if not (K == 3 and stats and (transposed or (stride == 1 and x.dtype == torch.bfloat16))):
            raise ValueError("dwconv3d(store=False) is the statistics-only mode of the K = 3 transposed / bf16 stride-1 convs")
        y = None
    elif y is None:
        y = torch.empty(oshape, dtype=x.dtype, device=x.device)
    st = None
    if stats:
        slots = nat.lib().pytc_dwconv3d_stat_slots(N, D, H, W, Cc, K, stride, dt, int(transposed))
        if slots < 0:
            raise RuntimeError(f"dwconv3d: unsupported channel count {Cc}")
        st = torch.empty((N, slots, 2, Cc), dtype=torch.float32, device=x.device)
    tag = f"C{Cc}_k{K}" + ("_s2" if stride == 2 and not transposed else "")
    if transposed:
        _run(f"dwconvT3d_fwd[{tag}]" if store else f"dwconvT3d_stats[{tag}]", _nbytes(x, y), nat.lib().pytc_dwconvT3d_fwd, _p(x), _p(y), _p(w_taps), _p(bias),
             _p(st), N, D, H, W, Cc, K, dt, _stream())
    else:
        sym = None
        if PROFILER.enabled:
            sym = _DW_VARIANTS.get(nat.lib().pytc_dwconv3d_kernel_variant(N, D, H, W, Cc, K, stride, dt, 0))
        fn = nat.lib().pytc_dwconv3d_fwd_wide if wide_range else nat.lib().pytc_dwconv3d_fwd
        _run(f"dwconv3d_fwd[{tag}]" if store else f"dwconv3d_stats[{tag}]", _nbytes(x, y), fn, _p(x), _p(y), _p(w_taps), _p(bias),
             _p(st), N, D, H, W, Cc, K, stride, dt, _stream(), symbol=sym)
    return y, st


_DW_VARIANTS = {0: "dwconv3d_direct_kernel", 1: "dwconv3d_k3_gather_kernel", 2: "dwconv3d_xblock_kernel",
                3: "dwconv3d_k3_march_kernel", 4: "dwconvT3d_k3_cell_kernel", 5: "dwconvT3d_kernel",
                6: "dwconv3d_k3_mfma_kernel", 7: "dwconvT3d_k3_tile_kernel", 8: "dwconv3d_k3_s2_march_kernel"}


def dwconv3d_res_supported(x: torch.Tensor, K: int, stride: int = 1) -> bool:
    N, D, H, W, Cc = x.shape
    return bool(nat.lib().pytc_dwconv3d_res_supported(D, H, W, Cc, K, stride, dtype_code(x.dtype)))


def dwconv3d_res(x: torch.Tensor, w_taps: torch.Tensor, res: torch.Tensor, *, K: int = 3) -> torch.Tensor:
    """y = dwconv3d(x, taps) + res in one kernel (bf16, z-march shapes: dwconv3d_res_supported); the data gradient of a residual
    block.  The residual is added to the fp32 accumulator: one rounding, where dwconv3d followed by add_ has two."""
    _dev(x, "x"); _dev(res, "res"); _dev(w_taps, "w_taps")
    if res.shape != x.shape or res.dtype != x.dtype:
        raise ValueError("dwconv3d_res: the residual must match the (stride-1) output shape and dtype")
    N, D, H, W, Cc = x.shape
    y = torch.empty_like(x)
    _run(f"dwconv3d_res[C{Cc}_k{K}]", _nbytes(x, y, res), nat.lib().pytc_dwconv3d_fwd_res, _p(x), _p(res), _p(y), _p(w_taps), None,
         N, D, H, W, Cc, K, 1, dtype_code(x.dtype), _stream(), symbol="dwconv3d_k3_march_kernel")
    return y


def layernorm_rows(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """Channels-first LayerNorm of MedNeXt: x (..., C) normalised over C per voxel row."""
    _dev(x, "x")
    Cc = x.shape[-1]
    y = torch.empty_like(x)
    _run(f"layernorm_rows[C{Cc}]", 2 * _nbytes(x), nat.lib().pytc_layernorm_rows, _p(x), _p(y), _p(gamma), _p(beta),
         x.numel() // Cc, Cc, float(eps), dtype_code(x.dtype), _stream())
    return y


def layernorm_rows_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: Optional[torch.Tensor], eps: float = 1e-5):
    """Backward of layernorm_rows: -> dx (like x), partial (slots, 2, C) fp32 whose sums over slots are dbeta ([:, 0]) and
    dgamma ([:, 1])."""
    _dev(dy, "dy"); _dev(x, "x")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    slots = nat.lib().pytc_layernorm_rows_bwd_slots(rows, Cc, dtype_code(x.dtype))
    if slots <= 0:
        raise ValueError(f"layernorm_rows_bwd: unsupported channel count {Cc}")
    dx = torch.empty_like(x)
    partial = torch.empty((slots, 2, Cc), dtype=torch.float32, device=x.device)
    _run(f"layernorm_rows_bwd[C{Cc}]", _nbytes(dy, x, dx), nat.lib().pytc_layernorm_rows_bwd, _p(dy), _p(x), _p(gamma), _p(dx),
         _p(partial), rows, Cc, float(eps), dtype_code(x.dtype), _stream())
    return dx, partial


def grn_bwd_apply(dh2: torch.Tensor, hp: torch.Tensor, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """(dh2 * A[n, c] + gelu(hp) * B[n, c]) * gelu'(hp) on (N, rows, C) tensors; A, B (N, C) fp32."""
    _dev(dh2, "dh2"); _dev(hp, "hp")
    N, Cc = hp.shape[0], hp.shape[-1]
    rows = hp.numel() // (N * Cc)
    out = torch.empty_like(hp)
    _run(f"grn_bwd_apply[C{Cc}]", _nbytes(dh2, hp, out), nat.lib().pytc_grn_bwd_apply, _p(dh2), _p(hp), _p(A), _p(B), _p(out), N,
         rows, Cc, dtype_code(hp.dtype), _stream())
    return out


def groupnorm_finalize(stats: torch.Tensor, count: float, gamma: Optional[torch.Tensor],
                       beta: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    N, slots, _, Cc = stats.shape
    ab = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    _run("groupnorm_finalize", _nbytes(stats, ab), nat.lib().pytc_groupnorm_finalize, _p(stats), slots, float(count),
         _p(gamma), _p(beta), float(eps), _p(ab), N, Cc, _stream())
    return ab


def groupnorm_fold_mlp_supported(c: int, c_hid: int) -> bool:
    return c in (32, 64, 128) and c_hid % 32 == 0 and 32 <= c_hid <= 512


def groupnorm_fold_mlp(stats: torch.Tensor, count: float, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float,
                       w2: torch.Tensor, b2: Optional[torch.Tensor], *, want_ab: bool = False):
    """GroupNorm finalize folded into the mixer's expanding conv (pytc_groupnorm_fold_mlp): stats (N, slots, 2, C), w2 fp32
    (C_hid, C) -> (w2n (N, C_hid * C) bf16 paired images, b2n (N, C_hid) fp32[, ab (N, 2, C)]); pw_mlp(t, None, w2n, b2n, ...) then
    equals pw_mlp(t, ab, pack(w2), b2, ...) up to the rounding point (the weight instead of the normalised activation)."""
    N, slots, _, Cc = stats.shape
    c_hid = int(w2.shape[0])
    if w2.dtype != torch.float32 or tuple(w2.shape) != (c_hid, Cc) or not w2.is_contiguous():
        raise ValueError(f"groupnorm_fold_mlp: w2 must be contiguous fp32 (C_hid, {Cc}), got {tuple(w2.shape)} {w2.dtype}")
    w2n = torch.empty((N, c_hid * Cc), dtype=torch.bfloat16, device=stats.device)
    b2n = torch.empty((N, c_hid), dtype=torch.float32, device=stats.device)
    ab = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device) if want_ab else None
    _run("groupnorm_finalize", _nbytes(stats, w2n, b2n), nat.lib().pytc_groupnorm_fold_mlp, _p(stats), slots, float(count),
         _p(gamma), _p(beta), float(eps), _p(w2), _p(b2), _p(w2n), _p(b2n), _p(ab), N, Cc, c_hid, _stream())
    return (w2n, b2n, ab) if want_ab else (w2n, b2n)


# ------------------------------------------------------------------ pointwise convs (MFMA GEMMs)
def pw_pack_weight(w: torch.Tensor, dtype: torch.dtype, *, transposed: bool = False) -> torch.Tensor:
    """w fp32 (C_out, C_in) [or (C_in, C_out) when transposed] -> packed MFMA operand image."""
    _dev(w, "w")
    if w.dtype != torch.float32 or w.dim() != 2:
        raise ValueError("pointwise weight must be float32 2-D")
    c_out, c_in = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    n = nat.lib().pytc_pw_packed_elems(c_out, c_in, dtype_code(dtype))
    packed = torch.empty((n,), dtype=dtype, device=w.device)
    _run("pw_pack_weight", _nbytes(w, packed), nat.lib().pytc_pw_pack_weight, _p(w), c_out, c_in, int(transposed),
         _p(packed), dtype_code(dtype), _stream())
    return packed


def pw_conv(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, N: int, rows_per_sample: int,
            c_in: int, c_out: int, out_dtype: torch.dtype, ab: Optional[torch.Tensor] = None,
            act: int = nat.ACT_NONE, res: Optional[torch.Tensor] = None, res_mode: int = nat.RES_NONE,
            gather: int = 0, grid: Sequence[int] = (0, 0, 0), res_low: Optional[torch.Tensor] = None,
            res_bias: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
            pre_act: int = nat.ACT_NONE, w_paired=False) -> torch.Tensor:
    _dev(x, "x"); _dev(w_packed, "w_packed")
    if y is None:
        y = torch.empty((N, rows_per_sample, c_out), dtype=out_dtype, device=x.device)
    a = nat.PwArgs()
    a.x, a.w_packed, a.bias, a.ab, a.res, a.y = (x.data_ptr(), w_packed.data_ptr(),
                                                 bias.data_ptr() if bias is not None else None,
                                                 ab.data_ptr() if ab is not None else None,
                                                 res.data_ptr() if res is not None else None, y.data_ptr())
    a.N, a.rows_per_sample, a.C_in, a.C_out = N, rows_per_sample, c_in, c_out
    a.in_dtype, a.out_dtype, a.w_dtype = dtype_code(x.dtype), dtype_code(y.dtype), dtype_code(w_packed.dtype)
    a.act, a.res_mode, a.gather, a.pre_act, a.w_paired = act, res_mode, gather, pre_act, int(w_paired)
    a.Di, a.Hi, a.Wi = (int(v) for v in grid)
    a.res_low = res_low.data_ptr() if res_low is not None else None
    a.res_bias = res_bias.data_ptr() if res_bias is not None else None
    rows_in = N * rows_per_sample * c_in * x.element_size()    # algorithmic: each operand once
    nb = rows_in + _nbytes(y) + (_nbytes(y) if res is not None else 0)
    _run(f"pw_conv_fwd[{c_in}->{c_out}]", nb, nat.lib().pytc_pw_conv_fwd, C.byref(a), _stream(),
         **({"symbol": "pw_gemm_lds_kernel"} if int(w_paired) == 2 else {}))
    return y


def pw_conv_paired_supported(*, c_in: int, c_out: int, in_dtype: torch.dtype, out_dtype: torch.dtype,
                             w_dtype: torch.dtype = torch.bfloat16, act: int = nat.ACT_NONE, gather: int = 0) -> bool:
    """True when pw_conv may be given a paired-row weight image (pw_pack_weight_paired) -> 16-byte-store kernel."""
    a = nat.PwArgs()
    a.C_in, a.C_out, a.act, a.gather = int(c_in), int(c_out), int(act), int(gather)
    try:
        a.in_dtype, a.out_dtype, a.w_dtype = dtype_code(in_dtype), dtype_code(out_dtype), dtype_code(w_dtype)
    except Exception:
        return False
    return bool(nat.lib().pytc_pw_conv_paired_supported(C.byref(a)))


def _mlp_symbol(a, plain: str, kind: int) -> str:
    """Device symbol of a fused-mixer launch for the profiler tables: `plain` (the one-tile-per-wave kernel's instance), or the DMA-prefetching
    kernel's instance when the library will pick it for these arguments (kind 0 plain / residual, 1 stem residual, 2 fused head)."""
    if not PROFILER.enabled:
        return plain
    if nat.lib().pytc_pw_mlp_dma_applies(C.byref(a), 1 if kind == 1 else 0):
        return f"pw_mlp_dma_kernel<{int(a.C_hid) // 32}, {kind}>"
    return plain


def pw_mlp_supported(c_in: int, c_hid: int, c_out: int) -> bool:
    return bool(nat.lib().pytc_pw_mlp_supported(int(c_in), int(c_hid), int(c_out)))


# The projecting GEMM of the fused mixers (w3) takes its weights as an fp16 image: the hidden activation is then produced by
# the packed-fp16 polynomial GELU (two values per VALU instruction, no transcendentals) and consumed by the f16 MFMA
# (csrc/pytc_common.h: gelu_h2).  False restores the bf16 image + sigmoid-form GELU (exp + rcp in fp32).
MLP_F16_PROJECT = True


def pw_pack_weight_paired(w: torch.Tensor, *, transposed: bool = False, f16: bool = False) -> torch.Tensor:
    """MFMA image with the paired-row permutation expected by pw_mlp (see csrc/pw_common.h): bf16, or fp16 (`f16=True`, the
    projection weights of a mixer that runs the packed-fp16 GELU; the tensor's dtype tells the launch wrappers which)."""
    _dev(w, "w")
    if w.dtype != torch.float32 or w.dim() != 2:
        raise ValueError("pointwise weight must be float32 2-D")
    c_out, c_in = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    n = nat.lib().pytc_pw_packed_elems(c_out, c_in, nat.BF16)
    packed = torch.empty((n,), dtype=torch.float16 if f16 else torch.bfloat16, device=w.device)
    fn = nat.lib().pytc_pw_pack_weight_paired_f16 if f16 else nat.lib().pytc_pw_pack_weight_paired
    _run("pw_pack_weight_paired", _nbytes(w, packed), fn, _p(w), c_out, c_in, int(transposed), _p(packed), _stream())
    return packed


PACK_PAIRED, PACK_PAIRED_T, PACK_PAIRED_F16, PACK_PAIRED_F16_T, PACK_TAPS, PACK_TAPS_FLIPPED, PACK_ROWMAJOR, PACK_ROWMAJOR_T = range(8)


class StepPacks:
    """The per-step weight re-layouts of ONE model (MFMA images of its 1x1x1 convs, tap-major depthwise stencils), rebuilt by
    a single launch per optimizer step (pytc_pack_multi) instead of one launch each.  Training code asks `lookup(w, kind)`;
    a miss is served by the individual pack function and `register`ed, so the set fills itself during the first step and
    `refresh()` -- called at the start of every training forward -- repacks everything whose parameter version moved.
    Entries hold a reference to their source weight, so a data pointer identifies a parameter for the life of the set."""

    def __init__(self):
        self.rows = []          # [src weight, out, kind, C_out, C_in, aux, n_elems, version]
        self.index = {}
        self.table = None
        self.total = 0

    def lookup(self, w: torch.Tensor, kind: int):
        i = self.index.get((w.data_ptr(), kind))
        if i is None:
            return None
        row = self.rows[i]
        return row[1] if row[7] == w._version else None

    def register(self, w: torch.Tensor, kind: int, out: torch.Tensor, c_out: int, c_in: int, aux: int) -> None:
        key = (w.data_ptr(), kind)
        i = self.index.get(key)
        row = [w, out, int(kind), int(c_out), int(c_in), int(aux), int(out.numel()), w._version]
        if i is None:
            self.index[key] = len(self.rows)
            self.rows.append(row)
        else:
            self.rows[i] = row
        self.table = None       # rebuilt (one small upload) at the next refresh

    def refresh(self) -> None:
        if not self.rows or all(r[7] == r[0]._version for r in self.rows):
            return
        dev = self.rows[0][1].device
        if self.table is None:
            flat, pos = [], 0
            for src, out, kind, c_out, c_in, aux, n, _ver in self.rows:
                flat += [src.data_ptr(), out.data_ptr(), kind, c_out, c_in, aux, pos, n]
                pos += n
            self.table = torch.tensor(flat, dtype=torch.int64).to(dev)
            self.total = pos
        _run("pack_multi", self.total * 6, nat.lib().pytc_pack_multi, _p(self.table), len(self.rows), self.total, _stream())
        for r in self.rows:
            r[7] = r[0]._version


def packed_paired(w_mat: torch.Tensor, *, transposed: bool = False, f16: bool = False, packs: Optional[StepPacks] = None):
    """pw_pack_weight_paired through a StepPacks set (w_mat: the fp32 (C_a, C_b) matrix view of the parameter)."""
    if packs is None:
        return pw_pack_weight_paired(w_mat, transposed=transposed, f16=f16)
    kind = (PACK_PAIRED_F16 if f16 else PACK_PAIRED) + (1 if transposed else 0)
    hit = packs.lookup(w_mat, kind)
    if hit is not None:
        return hit
    out = pw_pack_weight_paired(w_mat, transposed=transposed, f16=f16)
    c_out, c_in = (w_mat.shape[1], w_mat.shape[0]) if transposed else (w_mat.shape[0], w_mat.shape[1])
    packs.register(w_mat, kind, out, c_out, c_in, (c_in + 31) // 32)
    return out


def packed_rowmajor(w_mat: torch.Tensor, *, transposed: bool = False, packs: Optional[StepPacks] = None) -> torch.Tensor:
    """The plain row-major bf16 matrix [C_out][C_in] of a 1x1x1 conv weight (its transpose for a data gradient): the weights of
    pw_conv(w_paired=2), through a StepPacks set when given (one pack launch per step for the whole model)."""
    kind = PACK_ROWMAJOR_T if transposed else PACK_ROWMAJOR
    if packs is not None:
        hit = packs.lookup(w_mat, kind)
        if hit is not None:
            return hit
    out = (w_mat.t() if transposed else w_mat).to(torch.bfloat16).contiguous()
    if packs is not None:
        c_out, c_in = (w_mat.shape[1], w_mat.shape[0]) if transposed else (w_mat.shape[0], w_mat.shape[1])
        packs.register(w_mat, kind, out, c_out, c_in, 0)
    return out


def pw_conv_rowmajor_supported(*, c_in: int, c_out: int, in_dtype: torch.dtype, out_dtype: torch.dtype, act: int = nat.ACT_NONE,
                               gather: int = 0) -> bool:
    """True when pw_conv may run on the LDS-tiled GEMM with packed_rowmajor weights (w_paired=2)."""
    a = nat.PwArgs()
    a.C_in, a.C_out, a.act, a.gather = int(c_in), int(c_out), int(act), int(gather)
    try:
        a.in_dtype, a.out_dtype, a.w_dtype = dtype_code(in_dtype), dtype_code(out_dtype), dtype_code(torch.bfloat16)
    except Exception:
        return False
    return bool(nat.lib().pytc_pw_conv_rowmajor_supported(C.byref(a)))


def packed_taps(w: torch.Tensor, *, flipped: bool = False, packs: Optional[StepPacks] = None) -> torch.Tensor:
    """Depthwise conv weight (C, 1, k, k, k) fp32 -> tap-major (k^3, C) fp32 (reversed stencil when `flipped`)."""
    c, k3 = w.shape[0], w.shape[-1] ** 3
    src = w.detach()
    kind = PACK_TAPS_FLIPPED if flipped else PACK_TAPS
    if packs is not None and src.dtype == torch.float32 and src.is_contiguous():
        hit = packs.lookup(src, kind)
        if hit is not None:
            return hit
    out = src.float().reshape(c, k3).t().contiguous()
    if flipped:
        out = torch.flip(out, dims=[0]).contiguous()
    if packs is not None and src.dtype == torch.float32 and src.is_contiguous():
        packs.register(src, kind, out, c, 1, k3)
    return out


def _check_out(y: torch.Tensor, numel: int, dtype: torch.dtype, what: str) -> None:
    """A caller-provided output buffer (e.g. a sample slice of a larger batch tensor) must be a dense device tensor of the size
    and dtype the kernel writes."""
    _dev(y, "y")
    if y.dtype != dtype or y.numel() != numel or not y.is_contiguous():
        raise ValueError(f"{what}: output buffer must be a contiguous {dtype} tensor of {numel} elements, got "
                         f"{tuple(y.shape)} {y.dtype} contiguous={y.is_contiguous()}")


def _w3_format(w3p: torch.Tensor) -> int:
    return nat.W3_F16 if w3p.dtype == torch.float16 else nat.W3_BF16


def _folded_operands(ab, w2p: torch.Tensor, b2: torch.Tensor, N: int, c_in: int, c_hid: int) -> int:
    """ab is None <=> the caller passes groupnorm_fold_mlp's per-sample operands: shapes checked here, flag for pytc_mlp_args."""
    if ab is not None:
        return 0
    if tuple(w2p.shape) != (N, c_hid * c_in) or tuple(b2.shape) != (N, c_hid) or w2p.dtype != torch.bfloat16:
        raise ValueError(f"norm-folded mixer operands must be w2n (N, C_hid*C_in) bf16 and b2n (N, C_hid); got {tuple(w2p.shape)} "
                         f"{w2p.dtype} / {tuple(b2.shape)}")
    return 1


def pw_mlp(t: torch.Tensor, ab: Optional[torch.Tensor], w2p: torch.Tensor, b2: torch.Tensor, w3p: torch.Tensor,
           b3: torch.Tensor, *, N: int, rows_per_sample: int, c_in: int, c_hid: int, c_out: int,
           res: Optional[torch.Tensor] = None, res_mode: int = nat.RES_NONE, grid: Sequence[int] = (0, 0, 0),
           res_low: Optional[torch.Tensor] = None, res_bias: Optional[torch.Tensor] = None,
           y: Optional[torch.Tensor] = None, hidden_pre: Optional[torch.Tensor] = None, lds: bool = False,
           chunked: bool = False, train_nostore: bool = False) -> torch.Tensor:
    """Fused norm-apply -> 1x1 expand -> GELU -> 1x1 project (+residual) on bf16 NDHWC rows.  hidden_pre (N, rows, c_hid)
    bf16: training forward, the hidden pre-activation is stored there as well.  lds: the persistent kernel with the weight images
    resident in LDS (pw_mlp_lds_supported shapes, fp16 projection image; bit-identical).  chunked: the kernel whose workgroups stream
    the weight images through LDS one hidden chunk at a time (pw_mlp_chunk_supported shapes: wide hidden layers; bit-identical).
    train_nostore: the training forward's arithmetic (hidden pre-activation rounded to bf16 before the activation: y has the bits of the
    hidden_pre form) without the store -- for blocks whose backward rebuilds the hidden tensor (mixer_bwd_rc)."""
    _dev(t, "t")
    if t.dtype != torch.bfloat16:
        raise TypeError("pw_mlp runs on bfloat16 activations")
    if y is None:
        y = torch.empty((N, rows_per_sample, c_out), dtype=torch.bfloat16, device=t.device)
    else:
        _check_out(y, N * rows_per_sample * c_out, torch.bfloat16, "pw_mlp")
    a = nat.MlpArgs()
    a.t, a.ab, a.w2_packed, a.b2, a.w3_packed, a.b3 = (t.data_ptr(), None if ab is None else ab.data_ptr(), w2p.data_ptr(),
                                                       b2.data_ptr(), w3p.data_ptr(), b3.data_ptr())
    a.per_sample = _folded_operands(ab, w2p, b2, N, c_in, c_hid)
    a.w3_format = _w3_format(w3p)
    a.res = res.data_ptr() if res is not None else None
    a.res_low = res_low.data_ptr() if res_low is not None else None
    a.res_bias = res_bias.data_ptr() if res_bias is not None else None
    a.y = y.data_ptr()
    a.N, a.rows_per_sample, a.C_in, a.C_hid, a.C_out, a.res_mode = N, rows_per_sample, c_in, c_hid, c_out, res_mode
    a.Di, a.Hi, a.Wi = (int(v) for v in grid)
    nb = N * rows_per_sample * 2 * (c_in + c_out + (c_out if res is not None else 0))
    if hidden_pre is not None:
        _dev(hidden_pre, "hidden_pre")
        _run(f"pw_mlp_train_fwd[{c_in}->{c_hid}->{c_out}]", nb + N * rows_per_sample * 2 * c_hid, nat.lib().pytc_pw_mlp_train_fwd,
             C.byref(a), _p(hidden_pre), _stream())
        return y
    if train_nostore:
        _run(f"pw_mlp_train_fwd_nostore[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_train_fwd_nostore, C.byref(a), _stream(),
             symbol=f"pw_mlp_kernel<{c_in // 32}, {c_out // 16}>")
        return y
    if lds:
        _run(f"pw_mlp_fwd[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_lds_fwd, C.byref(a), _stream(),
             symbol=f"pw_mlp_lds_kernel<{c_in // 32}, {c_out // 16}>")
        return y
    if chunked:
        _run(f"pw_mlp_fwd[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_chunk_fwd, C.byref(a), _stream(),
             symbol=f"pw_mlp_chunk_kernel<{c_in // 32}, {c_out // 16}>")
        return y
    _run(f"pw_mlp_fwd[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_fwd, C.byref(a), _stream(),
         symbol=_mlp_symbol(a, f"pw_mlp_kernel<{c_in // 32}, {c_out // 16}>", 0))
    return y


def pw_mlp_lds_supported(c_in: int, c_hid: int, c_out: int) -> bool:
    return bool(nat.lib().pytc_pw_mlp_lds_supported(int(c_in), int(c_hid), int(c_out)))


def pw_mlp_chunk_supported(c_in: int, c_hid: int, c_out: int) -> bool:
    return bool(nat.lib().pytc_pw_mlp_chunk_supported(int(c_in), int(c_hid), int(c_out)))


def pw_gemm_supported(c_in: int, c_out: int) -> bool:
    return bool(nat.lib().pytc_pw_gemm_supported(int(c_in), int(c_out)))


def pw_gemm(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, *, N: int, rows_per_sample: int, ab: Optional[torch.Tensor] = None,
            gelu: bool = False, res: Optional[torch.Tensor] = None, res_mode: int = nat.RES_NONE, grid: Sequence[int] = (0, 0, 0),
            res_low: Optional[torch.Tensor] = None, res_bias: Optional[torch.Tensor] = None,
            y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One 1x1x1 conv of a deep-level block as an LDS-tiled MFMA GEMM (pytc_pw_gemm_fwd): x (N, rows, C_in) bf16 [+ GroupNorm affine
    ab] or fp16, w (C_out, C_in) row-major in x's dtype -> (N, rows, C_out): fp16 = GELU(acc) when `gelu`, else bf16 with the residual
    epilogue of pw_mlp."""
    _dev(x, "x"); _dev(w, "w")
    c_out, c_in = int(w.shape[0]), int(w.shape[1])
    if x.dtype not in (torch.bfloat16, torch.float16) or w.dtype != x.dtype or not w.is_contiguous() or int(x.shape[-1]) != c_in:
        raise TypeError(f"pw_gemm: x {tuple(x.shape)} {x.dtype} / w {tuple(w.shape)} {w.dtype}: same 16-bit type, w (C_out, C_in) contiguous")
    out_dt = torch.float16 if gelu else torch.bfloat16
    if y is None:
        y = torch.empty((N, rows_per_sample, c_out), dtype=out_dt, device=x.device)
    else:
        _check_out(y, N * rows_per_sample * c_out, out_dt, "pw_gemm")
    nb = N * rows_per_sample * 2 * (c_in + c_out + (c_out if res is not None else 0)) + 2 * c_in * c_out
    _run(f"pw_gemm[{c_in}->{c_out}]", nb, nat.lib().pytc_pw_gemm_fwd, _p(x), _p(w), _p(bias), _p(ab), _p(y), N, rows_per_sample, c_in,
         c_out, int(x.dtype == torch.float16), int(gelu), _p(res), _p(res_low), _p(res_bias), int(res_mode), *(int(v) for v in grid),
         _stream(), flops=2 * N * rows_per_sample * c_in * c_out, symbol="pw_gemm_lds_kernel")
    return y


def pw_mlp_up_supported(c_in: int, c_hid: int, c_out: int) -> bool:
    return bool(nat.lib().pytc_pw_mlp_up_supported(int(c_in), int(c_hid), int(c_out)))


def pw_mlp_up(x_low: torch.Tensor, taps: torch.Tensor, dw_bias: Optional[torch.Tensor], ab: torch.Tensor, w2p: torch.Tensor,
              b2: torch.Tensor, w3p: torch.Tensor, b3: torch.Tensor, skip: torch.Tensor, *, c_hid: int, c_out: int,
              res_low: Optional[torch.Tensor] = None, res_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused MedNeXt up block on bf16 NDHWC tensors: x_low (N,D,H,W,C_in) -> y (N,2D,2H,2W,C_out); the depthwise transposed
    conv output is formed in the mixer's prologue and never stored (csrc/pw_mlp_up_kernels.hip)."""
    _dev(x_low, "x_low"); _dev(skip, "skip"); _dev(taps, "taps")
    if x_low.dtype != torch.bfloat16 or skip.dtype != torch.bfloat16:
        raise TypeError("pw_mlp_up runs on bfloat16 activations")
    N, D, H, W, c_in = x_low.shape
    if tuple(skip.shape) != (N, 2 * D, 2 * H, 2 * W, c_out):
        raise ValueError(f"pw_mlp_up: skip shape {tuple(skip.shape)} does not match the up-sampled grid")
    y = torch.empty((N, 2 * D, 2 * H, 2 * W, c_out), dtype=torch.bfloat16, device=x_low.device)
    a = nat.MlpArgs()
    a.t, a.ab, a.w2_packed, a.b2, a.w3_packed, a.b3 = (x_low.data_ptr(), ab.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                                                       w3p.data_ptr(), b3.data_ptr())
    if w3p.dtype != torch.bfloat16:
        raise TypeError("pw_mlp_up takes the bf16 image of the projection weights")
    a.res = skip.data_ptr()
    if dw_bias is None:
        dw_bias = torch.zeros((c_in,), dtype=torch.float32, device=x_low.device)
    if res_bias is None:
        res_bias = torch.zeros((c_out,), dtype=torch.float32, device=x_low.device)
    a.res_low = res_low.data_ptr() if res_low is not None else None
    a.res_bias = res_bias.data_ptr()
    a.y = y.data_ptr()
    a.N, a.rows_per_sample, a.C_in, a.C_hid, a.C_out, a.res_mode = N, 8 * D * H * W, c_in, c_hid, c_out, nat.RES_UPSAMPLE
    a.Di, a.Hi, a.Wi = D, H, W
    nb = _nbytes(x_low, skip, y) + (_nbytes(res_low) if res_low is not None else 0)
    _run(f"pw_mlp_up_fwd[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_up_fwd, C.byref(a), _p(taps), _p(dw_bias),
         _stream())
    return y


def stem_dwconv3d_supported(c_in: int, c: int, K: int) -> bool:
    return bool(nat.lib().pytc_stem_dwconv3d_supported(int(c_in), int(c), int(K)))


def stem_dwconv3d_pack(stem_w: torch.Tensor, stem_b: torch.Tensor, w_taps: torch.Tensor, bias: Optional[torch.Tensor]):
    """-> (wx (27,C), wb (27,C), cst (C), image): the fp32 products pytc_stem_dwconv3d_fwd consumes and the f16 fragment image of
    its matrix-core form (pytc_stem_dwconv3d_pack_mfma).  Cache them per weight version."""
    wx = (w_taps * stem_w.view(1, -1)).contiguous()
    wb = (w_taps * stem_b.view(1, -1)).contiguous()
    cst = (wb.sum(0) + (bias if bias is not None else 0.0)).contiguous()
    image = torch.empty((nat.lib().pytc_stem_dwconv3d_mfma_image_bytes(),), dtype=torch.uint8, device=wx.device)
    _run("stem_dwconv3d_pack_mfma", 0, nat.lib().pytc_stem_dwconv3d_pack_mfma, _p(wx), _p(wb), _p(cst), _p(image), _stream())
    return wx, wb, cst, image


def stem_dwconv3d(x: torch.Tensor, packed):
    """x (N,D,H,W,1) fp32 -> (t (N,D,H,W,32) bf16 = dwconv3(stem(x)), stats (N,slots,2,32)); the stem output is not formed."""
    wx, wb, cst, image = packed
    _dev(x, "x"); _dev(wx, "wx")
    if x.dtype != torch.float32 or x.shape[-1] != 1:
        raise TypeError("stem_dwconv3d takes the 1-channel fp32 network input")
    N, D, H, W, _ = x.shape
    Cc = wx.shape[1]
    y = torch.empty((N, D, H, W, Cc), dtype=torch.bfloat16, device=x.device)
    slots = nat.lib().pytc_stem_dwconv3d_stat_slots(D, H, W)
    st = torch.empty((N, slots, 2, Cc), dtype=torch.float32, device=x.device)
    _run(f"stem_dwconv3d_fwd[C{Cc}_k3]", _nbytes(x, y), nat.lib().pytc_stem_dwconv3d_fwd, _p(x), _p(wx), _p(wb), _p(cst),
         _p(image), _p(y), _p(st), N, D, H, W, Cc, _stream())
    return y, st


def pw_mlp_stemres(t: torch.Tensor, ab: torch.Tensor, w2p: torch.Tensor, b2: torch.Tensor, w3p: torch.Tensor, b3: torch.Tensor,
                   x0: torch.Tensor, stem_w: torch.Tensor, stem_b: torch.Tensor, *, N: int, rows_per_sample: int, c_in: int,
                   c_hid: int, c_out: int, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pw_mlp whose residual is the stem output recomputed from the 1-channel fp32 input x0 (N, rows)."""
    _dev(t, "t"); _dev(x0, "x0")
    if t.dtype != torch.bfloat16 or x0.dtype != torch.float32:
        raise TypeError("pw_mlp_stemres: bf16 activations and the fp32 network input")
    if y is None:
        y = torch.empty((N, rows_per_sample, c_out), dtype=torch.bfloat16, device=t.device)
    else:
        _check_out(y, N * rows_per_sample * c_out, torch.bfloat16, "pw_mlp_stemres")
    a = nat.MlpArgs()
    a.t, a.ab, a.w2_packed, a.b2, a.w3_packed, a.b3 = (t.data_ptr(), None if ab is None else ab.data_ptr(), w2p.data_ptr(),
                                                       b2.data_ptr(), w3p.data_ptr(), b3.data_ptr())
    a.per_sample = _folded_operands(ab, w2p, b2, N, c_in, c_hid)
    a.w3_format = _w3_format(w3p)
    a.res = a.res_low = a.res_bias = None
    a.y = y.data_ptr()
    a.N, a.rows_per_sample, a.C_in, a.C_hid, a.C_out, a.res_mode = N, rows_per_sample, c_in, c_hid, c_out, nat.RES_ADD
    a.Di = a.Hi = a.Wi = 0
    nb = N * rows_per_sample * (2 * (c_in + c_out) + 4)
    _run(f"pw_mlp_fwd[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_stemres_fwd, C.byref(a), _p(x0), _p(stem_w),
         _p(stem_b), _stream(), symbol=_mlp_symbol(a, f"pw_mlp_kernel<{c_in // 32}, {c_out // 16}>+stemres", 1))
    return y


_MLP_BWD_CONSTS: dict = {}


def pw_mlp_bwd(dy: torch.Tensor, hidden_pre: torch.Tensor, w3t_p: torch.Tensor, w2t_p: torch.Tensor, *, N: int,
               rows_per_sample: int, c_in: int, c_hid: int, c_out: int):
    """Data gradients of a block's mixer in one launch.  dy (N, rows, c_out) bf16 -> (dx (N, rows, c_in), d_hidden
    (N, rows, c_hid)); w3t_p / w2t_p = pw_pack_weight_paired(W3 / W2, transposed=True)."""
    _dev(dy, "dy"); _dev(hidden_pre, "hidden_pre")
    if dy.dtype != torch.bfloat16:
        raise TypeError("pw_mlp_bwd runs on bfloat16 activations")
    dev = dy.device
    dx = torch.empty((N, rows_per_sample, c_in), dtype=torch.bfloat16, device=dev)
    dh = torch.empty((N, rows_per_sample, c_hid), dtype=torch.bfloat16, device=dev)
    key = (dev, N, c_in, c_hid, c_out)
    consts = _MLP_BWD_CONSTS.get(key)
    if consts is None:           # identity affine + zero biases: constants, made once per shape
        consts = (torch.cat([torch.ones((N, 1, c_out), device=dev), torch.zeros((N, 1, c_out), device=dev)], 1).contiguous(),
                  torch.zeros((c_hid,), device=dev), torch.zeros((c_in,), device=dev))
        _MLP_BWD_CONSTS[key] = consts
    ident, zh, zi = consts
    a = nat.MlpArgs()
    a.t, a.ab, a.w2_packed, a.b2, a.w3_packed, a.b3 = (dy.data_ptr(), ident.data_ptr(), w3t_p.data_ptr(), zh.data_ptr(),
                                                       w2t_p.data_ptr(), zi.data_ptr())
    a.w3_format = nat.W3_BF16
    a.res = a.res_low = a.res_bias = None
    a.y = dx.data_ptr()
    a.N, a.rows_per_sample, a.C_in, a.C_hid, a.C_out, a.res_mode = N, rows_per_sample, c_out, c_hid, c_in, nat.RES_NONE
    a.Di = a.Hi = a.Wi = 0
    nb = N * rows_per_sample * 2 * (c_out + 2 * c_hid + c_in)
    _run(f"pw_mlp_bwd[{c_out}->{c_hid}->{c_in}]", nb, nat.lib().pytc_pw_mlp_bwd, C.byref(a), _p(hidden_pre), _p(dh), _stream())
    return dx, dh


def pw_mlp_proj_supported(c_in: int, c_hid: int, c_out: int, c_proj: int) -> bool:
    return bool(nat.lib().pytc_pw_mlp_proj_supported(int(c_in), int(c_hid), int(c_out), int(c_proj)))


def pw_mlp_proj(t: torch.Tensor, w2n: torch.Tensor, b2n: torch.Tensor, w3p: torch.Tensor, b3: torch.Tensor, proj_w: torch.Tensor,
                proj_b: Optional[torch.Tensor], *, N: int, rows_per_sample: int, c_in: int, c_hid: int, c_out: int,
                res: Optional[torch.Tensor] = None, store_y: bool = False):
    """pw_mlp of a 32-channel block (norm-folded operands) with a 32 -> 32 conv of its output in the epilogue -> (y | None, z (N, rows, 32)
    bf16 = bf16(W bf16(y) + b)).  proj_w: pw_pack_weight_paired of the conv's (32, 32) weight."""
    _dev(t, "t"); _dev(proj_w, "proj_w")
    if t.dtype != torch.bfloat16 or proj_w.dtype != torch.bfloat16 or proj_w.numel() != 32 * 32:
        raise TypeError("pw_mlp_proj runs on bfloat16 activations and the paired bf16 image of a 32 x 32 projection")
    y = torch.empty((N, rows_per_sample, c_out), dtype=torch.bfloat16, device=t.device) if store_y else None
    z = torch.empty((N, rows_per_sample, 32), dtype=torch.bfloat16, device=t.device)
    a = nat.MlpArgs()
    a.t, a.ab, a.w2_packed, a.b2, a.w3_packed, a.b3 = t.data_ptr(), None, w2n.data_ptr(), b2n.data_ptr(), w3p.data_ptr(), b3.data_ptr()
    a.per_sample = _folded_operands(None, w2n, b2n, N, c_in, c_hid)
    a.w3_format = _w3_format(w3p)
    a.res = res.data_ptr() if res is not None else None
    a.res_low = a.res_bias = None
    a.y = y.data_ptr() if y is not None else None
    a.N, a.rows_per_sample, a.C_in, a.C_hid, a.C_out = N, rows_per_sample, c_in, c_hid, c_out
    a.res_mode = nat.RES_ADD if res is not None else nat.RES_NONE
    a.Di = a.Hi = a.Wi = 0
    nb = N * rows_per_sample * 2 * (c_in + (c_out if res is not None else 0) + (c_out if store_y else 0) + 32)
    _run(f"pw_mlp_fwd[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_proj_fwd, C.byref(a), _p(proj_w), _p(proj_b), _p(z),
         int(store_y), _stream(), symbol=f"pw_mlp_dma_kernel<{c_hid // 32}, 3>")
    return y, z


def pw_mlp_head_supported(c_in: int, c_hid: int, c_out: int) -> bool:
    return bool(nat.lib().pytc_pw_mlp_head_supported(int(c_in), int(c_hid), int(c_out)))


def pack_head_fragment(w: torch.Tensor) -> torch.Tensor:
    """w (n_head <= 16, 32) fp32 -> (n_head, 64, 8) bf16 whose [0] is the 16x32 MFMA A-fragment image (the leading
    dimension only carries n_head to pw_mlp_head)."""
    n_head = w.shape[0]
    if n_head > 16 or w.shape[1] != 32:
        raise ValueError("head weights must be (n_head <= 16, 32)")
    w16 = torch.zeros((16, 32), dtype=torch.float32, device=w.device)
    w16[:n_head] = w
    frag = w16.view(16, 4, 8).permute(1, 0, 2).reshape(64, 8).to(torch.bfloat16)
    return frag.unsqueeze(0).expand(n_head, 64, 8).contiguous()


def pw_mlp_head(t: torch.Tensor, ab: torch.Tensor, w2p: torch.Tensor, b2: torch.Tensor, w3p: torch.Tensor, b3: torch.Tensor,
                head_w: torch.Tensor, head_b: Optional[torch.Tensor], *, N: int, rows_per_sample: int, c_in: int,
                c_hid: int, c_out: int, res: Optional[torch.Tensor] = None, store_y: bool = False):
    """pw_mlp of the last block with the output projection in its epilogue -> (y | None, logits (N, rows, n_head) fp32).
    head_w: (n_head, 64, 8) bf16 fragment image from pack_head_fragment."""
    _dev(t, "t"); _dev(head_w, "head_w")
    if t.dtype != torch.bfloat16 or head_w.dtype != torch.bfloat16 or tuple(head_w.shape[1:]) != (64, 8):
        raise TypeError("pw_mlp_head runs on bfloat16 activations and a bf16 head fragment image (n_head, 64, 8)")
    n_head = head_w.shape[0]
    y = torch.empty((N, rows_per_sample, c_out), dtype=torch.bfloat16, device=t.device) if store_y else None
    logits = torch.empty((N, rows_per_sample, n_head), dtype=torch.float32, device=t.device)
    a = nat.MlpArgs()
    a.t, a.ab, a.w2_packed, a.b2, a.w3_packed, a.b3 = (t.data_ptr(), None if ab is None else ab.data_ptr(), w2p.data_ptr(),
                                                       b2.data_ptr(), w3p.data_ptr(), b3.data_ptr())
    a.per_sample = _folded_operands(ab, w2p, b2, N, c_in, c_hid)
    a.w3_format = _w3_format(w3p)
    a.res = res.data_ptr() if res is not None else None
    a.res_low = a.res_bias = None
    a.y = y.data_ptr() if y is not None else None
    a.N, a.rows_per_sample, a.C_in, a.C_hid, a.C_out = N, rows_per_sample, c_in, c_hid, c_out
    a.res_mode = nat.RES_ADD if res is not None else nat.RES_NONE
    a.Di = a.Hi = a.Wi = 0
    nb = N * rows_per_sample * (2 * (c_in + (c_out if res is not None else 0) + (c_out if store_y else 0)) + 4 * n_head)
    # same kernel template and GEMM shape as pw_mlp (HEAD flag): one label, each launch with its own byte count
    _run(f"pw_mlp_fwd[{c_in}->{c_hid}->{c_out}]", nb, nat.lib().pytc_pw_mlp_head_fwd, C.byref(a), _p(head_w),
         _p(head_b), _p(logits), n_head, int(store_y), _stream(),
         symbol=_mlp_symbol(a, f"pw_mlp_kernel<{c_in // 32}, {c_out // 16}>+head", 2))
    return y, logits


def dwmix_supported(x: torch.Tensor, c_hid: int, c_out: int) -> bool:
    """pytc_dwmix_fwd covers this block input: bf16 (N, D, H, W, 32), hidden width 64 / 96 / 128, 32 output channels, a shape of the
    matrix-core depthwise kernel."""
    if x.dim() != 5 or x.dtype != torch.bfloat16:
        return False
    _, D, H, W, Cc = x.shape
    return bool(nat.lib().pytc_dwmix_supported(D, H, W, Cc, int(c_hid), int(c_out), dtype_code(x.dtype)))


def dwmix(x: torch.Tensor, w_taps: torch.Tensor, dw_bias: Optional[torch.Tensor], w2n: torch.Tensor, b2n: torch.Tensor,
          w3p: torch.Tensor, b3: torch.Tensor, *, c_hid: int, c_out: int = 32, residual: bool = True, y: Optional[torch.Tensor] = None,
          head_w: Optional[torch.Tensor] = None, head_b: Optional[torch.Tensor] = None, store_y: bool = True):
    """Fused MedNeXt residual block (pytc_dwmix_fwd): depthwise 3x3x3 conv re-formed in LDS -> folded norm -> expand -> GELU -> project
    -> + x, the depthwise tensor never stored.  (w2n, b2n) = groupnorm_fold_mlp(statistics of dwconv3d(x, ..., store=False)).
    -> y (N, D, H, W, 32) bf16, or with head_w (pack_head_fragment): (y | None, logits (N, D, H, W, n_head) fp32)."""
    _dev(x, "x"); _dev(w_taps, "w_taps"); _dev(w2n, "w2n"); _dev(w3p, "w3p")
    N, D, H, W, Cc = x.shape
    if x.dtype != torch.bfloat16 or w2n.dtype != torch.bfloat16 or w3p.dtype != torch.float16:
        raise TypeError("dwmix runs on bfloat16 activations, per-sample bf16 expand images and the fp16 projection image")
    if tuple(w2n.shape) != (N, c_hid * Cc) or tuple(b2n.shape) != (N, c_hid) or tuple(w_taps.shape) != (27, Cc):
        raise ValueError(f"dwmix: operands do not match N={N}, C={Cc}, C_hid={c_hid}: w2n {tuple(w2n.shape)}, b2n {tuple(b2n.shape)}, taps {tuple(w_taps.shape)}")
    logits = None
    n_head = 0
    if head_w is not None:
        if head_w.dtype != torch.bfloat16 or tuple(head_w.shape[1:]) != (64, 8):
            raise TypeError("dwmix: the head fragment image is (n_head, 64, 8) bf16 (pack_head_fragment)")
        n_head = int(head_w.shape[0])
        logits = torch.empty((N, D, H, W, n_head), dtype=torch.float32, device=x.device)
        if not store_y:
            y = None
    if y is None and (head_w is None or store_y):
        y = torch.empty((N, D, H, W, c_out), dtype=torch.bfloat16, device=x.device)
    if y is not None and (tuple(y.shape) != (N, D, H, W, c_out) or y.dtype != torch.bfloat16 or not y.is_contiguous()):
        raise ValueError(f"dwmix: output buffer must be contiguous bf16 {(N, D, H, W, c_out)}")
    nb = _nbytes(x) * (2 if residual else 1) + _nbytes(y, logits)          # x read for the conv and (from L2) as the residual
    _run(f"dwmix_fwd[{Cc}->{c_hid}->{c_out}]", nb, nat.lib().pytc_dwmix_fwd, _p(x), _p(w_taps), _p(dw_bias), _p(w2n), _p(b2n), _p(w3p),
         _p(b3), int(bool(residual)), _p(y), _p(head_w), _p(head_b), _p(logits), n_head, N, D, H, W, Cc, int(c_hid), int(c_out),
         dtype_code(x.dtype), _stream(), symbol="dwconv3d_k3_mfma_kernel+mix" + ("+head" if head_w is not None else ""))
    return y if head_w is None else (y, logits)


# ------------------------------------------------------------------ dense conv / norm / pool (RSUNet)
def conv3d_pack_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """w fp32 (C_out, C_in, kd, kh, kw) -> packed MFMA image."""
    _dev(w, "w")
    co, ci, kd, kh, kw = w.shape
    n = nat.lib().pytc_conv3d_packed_elems(co, ci, kd, kh, kw, dtype_code(dtype))
    packed = torch.empty((n,), dtype=dtype, device=w.device)
    _run("conv3d_pack_weight", _nbytes(w, packed), nat.lib().pytc_conv3d_pack_weight, _p(w), co, ci, kd, kh, kw,
         _p(packed), dtype_code(dtype), _stream())
    return packed


def conv3d_pack_weight_dgrad(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """forward weight fp32 (C_out, C_in, kd, kh, kw) -> packed image of the data-gradient conv (C_out -> C_in, mirrored taps)."""
    _dev(w, "w")
    co, ci, kd, kh, kw = w.shape
    n = nat.lib().pytc_conv3d_packed_elems(ci, co, kd, kh, kw, dtype_code(dtype))
    packed = torch.empty((n,), dtype=dtype, device=w.device)
    _run("conv3d_pack_weight_dgrad", _nbytes(w, packed), nat.lib().pytc_conv3d_pack_weight_dgrad, _p(w), co, ci, kd, kh, kw,
         _p(packed), dtype_code(dtype), _stream())
    return packed


# layouts of a conv weight image: (pack function, channel / stride rule).  "fwd" / "dgrad": the stride-1 'same' conv and its data
# gradient (conv3d_pack_weight / _dgrad: LDS-tiled flat layout when the channel count allows); the other four: conv3d_strided
_CONV_LAYOUTS = ("fwd", "dgrad", "conv", "convT", "conv_dgrad", "convT_dgrad")


def _conv_layout_rule(shape, layout: str):
    """-> (C_out, C_in of the conv the image is for, s_o, s_c, flip, direct) for a 5-D weight of this shape"""
    a, b, kd, kh, kw = (int(v) for v in shape)
    ntap = kd * kh * kw
    if layout == "fwd":
        return a, b, b * ntap, ntap, 0, 0
    if layout == "dgrad":
        return b, a, ntap, b * ntap, 1, 0
    if layout in ("conv", "convT_dgrad"):
        return a, b, b * ntap, ntap, 0, 1
    if layout in ("convT", "conv_dgrad", "convT_phase", "conv_dgrad_phase"):
        return b, a, ntap, b * ntap, 0, 1
    raise ValueError(f"unknown weight layout {layout!r}")


_PHASE_LAYOUTS = ("convT_phase", "conv_dgrad_phase")


def convT3d_phase_plan(c_out: int, c_in: int):
    """(offsets[8], total elements, KC, chunks, groups[8]) of the eight phase images of a stride-2 transposed gather, or None when the
    shape has no LDS-tile plan (pytc_convT3d_phase_plan)."""
    out = (C.c_int64 * 19)()
    if nat.lib().pytc_convT3d_phase_plan(int(c_out), int(c_in), nat.BF16, out) != nat.OK:
        return None
    v = [int(x) for x in out]
    return v[0:8], v[8], v[9], v[10], v[11:19]


def _conv_pack_row(w: torch.Tensor, out_ptr: int, layout: str, dtype: torch.dtype, pad_to, first_block: int):
    """One row of pytc_conv3d_pack_multi's table for weight `w` (fp32, contiguous): the image is laid out for (co_p, ci_p) = pad_to
    channels (default: the conv's own), the SOURCE bounds stay the weight's -- the kernel writes zeros beyond them.  -> (row, n)"""
    co, ci, s_o, s_c, flip, direct = _conv_layout_rule(w.shape, layout)
    co_p, ci_p = (co, ci) if pad_to is None else pad_to
    if co_p < co or ci_p < ci:
        raise ValueError(f"padded conv image ({co_p}, {ci_p}) smaller than the weight's ({co}, {ci})")
    kd, kh, kw = (int(v) for v in w.shape[2:])
    if layout in _PHASE_LAYOUTS:
        # eight rows of kind 2 (one per output parity) into ONE buffer: the images of pytc_convT3d_phase_fwd
        plan = convT3d_phase_plan(co_p, ci_p)
        if plan is None or (kd, kh, kw) != (3, 3, 3) or dtype != torch.bfloat16 or pad_to is not None:
            raise ValueError(f"no phase images for a {tuple(w.shape)} weight in {dtype}")
        offs, total, kc, nchunks, groups = plan
        rows, blk = [], first_block
        for ph in range(8):
            n_ph = (offs[ph + 1] if ph < 7 else total) - offs[ph]
            ntap_ph = (1 + ((ph >> 2) & 1)) * (1 + ((ph >> 1) & 1)) * (1 + (ph & 1))
            rows += [w.data_ptr(), out_ptr + 2 * offs[ph], s_o, s_c, blk, n_ph, co, ci, ntap_ph, 2, 0, ph, kc, nchunks, groups[ph], 0]
            blk += (n_ph + 255) // 256
        return rows, total
    plan = (C.c_int64 * 5)()
    nat.check(nat.lib().pytc_conv3d_pack_plan(co_p, ci_p, kd, kh, kw, dtype_code(dtype), direct, plan), "conv3d_pack_plan")
    kind, p1, p2, p3, n = (int(v) for v in plan)
    return [w.data_ptr(), out_ptr, s_o, s_c, first_block, n, co, ci, kd * kh * kw, kind, int(dtype == torch.float32), flip,
            p1, p2, p3, 0], n


def _row_blocks(row) -> int:
    """256-thread blocks of a (possibly multi-row) table entry: sum over its rows of ceil(elements / 256)"""
    return sum((row[i + 5] + 255) // 256 for i in range(0, len(row), 16))


def conv3d_pack_weight_padded(w: torch.Tensor, layout: str, dtype: torch.dtype, pad_to) -> torch.Tensor:
    """The image of `w` for a conv that runs with pad_to = (C_out, C_in) channels (pad_channels): one-row pytc_conv3d_pack_multi."""
    _dev(w, "w")
    co, ci, *_ = _conv_layout_rule(w.shape, layout)
    kd, kh, kw = (int(v) for v in w.shape[2:])
    direct = _conv_layout_rule(w.shape, layout)[5]
    n_alloc = (nat.lib().pytc_conv3d_packed_elems(pad_to[0], pad_to[1], kd, kh, kw, dtype_code(dtype)) if not direct
               else None)
    if n_alloc is None:
        raise NotImplementedError("padded images exist for the stride-1 conv layouts ('fwd', 'dgrad')")
    out = torch.zeros((n_alloc,), dtype=dtype, device=w.device)
    row, n = _conv_pack_row(w, out.data_ptr(), layout, dtype, pad_to, 0)
    table = torch.tensor(row, dtype=torch.int64).to(w.device)
    _run("conv3d_pack_multi", _nbytes(w, out), nat.lib().pytc_conv3d_pack_multi, _p(table), 1, (n + 255) // 256, _stream())
    return out


class ConvPackSet:
    """The per-step conv-weight images of the dense-conv models (RSUNet, MONAI-style U-Net) in training: every image asked for
    through `get` is remembered, and `refresh()` -- called at the start of a training forward -- rebuilds ALL images whose
    weight changed since in ONE launch (pytc_conv3d_pack_multi) instead of one launch per conv and direction (64 per RSUNet
    step).  Images are bit-identical to the single packs.  Rows hold their weight weakly: a dropped model drops its rows."""

    def __init__(self):
        self.rows = {}          # (data_ptr, layout, dtype) -> [weakref(weight), out, version, plan tuple]
        self.table = None
        self.blocks = 0
        self.n_rows = 0

    def get(self, weight: torch.Tensor, layout: str, dtype: torch.dtype, pad_to=None) -> torch.Tensor:
        """pad_to = (C_out, C_in) the conv RUNS with (pad_channels; in the orientation of the image, i.e. already swapped for
        'dgrad'), None = the weight's own."""
        if pad_to is not None:
            own = _conv_layout_rule(weight.shape, layout)[:2]
            pad_to = None if tuple(pad_to) == tuple(own) else (int(pad_to[0]), int(pad_to[1]))
        key = (weight.data_ptr(), layout, dtype, pad_to)
        row = self.rows.get(key)
        if row is not None and row[0]() is weight and row[2] == weight._version:
            return row[1]
        w32 = weight.detach()
        if w32.dtype != torch.float32 or not w32.is_contiguous():       # not a plain fp32 parameter: single pack, not tracked
            return _conv_pack_single(w32.float().contiguous(), layout, dtype, pad_to)
        out = _conv_pack_single(w32, layout, dtype, pad_to)
        import weakref
        self.rows[key] = [weakref.ref(weight), out, weight._version, None]
        self.table = None
        return out

    def refresh(self) -> None:
        dead = [k for k, r in self.rows.items() if r[0]() is None]
        for k in dead:
            del self.rows[k]
        if dead:
            self.table = None
        live = [(k, r, r[0]()) for k, r in self.rows.items()]
        if not live or all(r[2] == w._version for _k, r, w in live):
            return
        dev = live[0][1][1].device
        if self.table is None:
            flat, blk = [], 0
            for (_ptr, layout, dtype, pad_to), r, w in live:
                row, _n = _conv_pack_row(w, r[1].data_ptr(), layout, dtype, pad_to, blk)
                flat += row
                blk += _row_blocks(row)
            self.table = torch.tensor(flat, dtype=torch.int64).to(dev)
            self.blocks = blk
            self.n_rows = len(flat) // 16
        _run("conv3d_pack_multi", sum(2 * r[1].numel() * r[1].element_size() for _k, r, _w in live), nat.lib().pytc_conv3d_pack_multi,
             _p(self.table), self.n_rows, self.blocks, _stream())
        for _k, r, w in live:
            r[2] = w._version


def _conv_pack_single(w32: torch.Tensor, layout: str, dtype: torch.dtype, pad_to=None) -> torch.Tensor:
    if pad_to is not None:
        return conv3d_pack_weight_padded(w32, layout, dtype, pad_to)
    if layout == "fwd":
        return conv3d_pack_weight(w32, dtype)
    if layout == "dgrad":
        return conv3d_pack_weight_dgrad(w32, dtype)
    if layout in _PHASE_LAYOUTS:
        _dev(w32, "w")
        co, ci, *_ = _conv_layout_rule(w32.shape, layout)
        plan = convT3d_phase_plan(co, ci)
        if plan is None:
            raise ValueError(f"no phase images for a {tuple(w32.shape)} weight")
        out = torch.empty((plan[1],), dtype=dtype, device=w32.device)
        rows, _n = _conv_pack_row(w32, out.data_ptr(), layout, dtype, None, 0)
        table = torch.tensor(rows, dtype=torch.int64).to(w32.device)
        _run("conv3d_pack_multi", _nbytes(w32, out), nat.lib().pytc_conv3d_pack_multi, _p(table), 8, _row_blocks(rows), _stream())
        return out
    return conv3d_pack_weight_direct(w32, dtype, layout=layout)


CONV_PACKS = ConvPackSet()


def norm_bwd_means(s: torch.Tensor, gamma: Optional[torch.Tensor], groups: int, rows: int, *, want_gamma: bool,
                   want_beta: bool, cpg: int = 0):
    """s (N,2,C) -> (M (N,2,C), dgamma (C) | None, dbeta (C) | None); groups = 0: batch statistics; cpg > 0: `groups` groups of
    `cpg` real channels, the rest of C is alignment padding (M = 0 there; gamma holds groups * cpg entries)"""
    _dev(s, "s")
    N, _, Cc = s.shape
    M = torch.empty_like(s)
    dg = torch.empty((Cc,), dtype=torch.float32, device=s.device) if want_gamma else None
    db = torch.empty((Cc,), dtype=torch.float32, device=s.device) if want_beta else None
    _run("norm_bwd_means", 2 * _nbytes(s), nat.lib().pytc_norm_bwd_means_cpg, _p(s), _p(gamma), _p(M), _p(dg), _p(db), N, Cc,
         int(groups), int(cpg), float(rows), _stream())
    return M, dg, db


def bn_train_finalize(stats: torch.Tensor, count: float, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float,
                      momentum: float, running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
                      num_batches_tracked: Optional[torch.Tensor], N: int):
    """BatchNorm (training) after the statistics pass in one launch: stats (N, slots, 2, C) of the whole batch -> ab (N,2,C),
    mean_rstd (N,2,C); blends the running buffers and increments num_batches_tracked in place (pass None to skip either)."""
    _dev(stats, "stats")
    Cc = stats.shape[-1]
    slots_total = stats.numel() // (2 * Cc)
    ab = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    mr = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    if running_mean is not None and (running_mean.dtype != torch.float32 or running_var.dtype != torch.float32):
        raise RuntimeError("bn_train_finalize: fp32 running buffers expected")
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        raise RuntimeError("bn_train_finalize: int64 num_batches_tracked expected")
    # channels beyond the norm's own count (gamma / running buffers) are alignment padding of the activation (pad_channels)
    c_real = int(gamma.numel()) if gamma is not None else (int(running_mean.numel()) if running_mean is not None else Cc)
    _run("bn_train_finalize", _nbytes(stats, ab, mr), nat.lib().pytc_bn_train_finalize_cpad, _p(stats), slots_total, float(count),
         _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(ab), _p(mr),
         N, Cc, c_real, _stream())
    return ab, mr


def bn_update_running(mean_rstd: torch.Tensor, running_mean: torch.Tensor, running_var: torch.Tensor, count: float,
                      eps: float, momentum: float) -> None:
    _dev(mean_rstd, "mean_rstd"); _dev(running_mean, "running_mean"); _dev(running_var, "running_var")
    if running_mean.dtype != torch.float32 or running_var.dtype != torch.float32:
        raise RuntimeError("bn_update_running: fp32 running buffers expected")
    _run("bn_update_running", 0, nat.lib().pytc_bn_update_running, _p(mean_rstd), _p(running_mean), _p(running_var),
         running_mean.numel(), float(count), float(eps), float(momentum), _stream())


def conv3d(x: torch.Tensor, w_packed: torch.Tensor, *, c_out: int, kernel, bias: Optional[torch.Tensor] = None,
           ab: Optional[torch.Tensor] = None, act_in: int = nat.ACT_NONE, act_param: float = 0.0,
           res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (N,D,H,W,C_in) -> (N,D,H,W,C_out): y = W * act_in(a*x+b) + bias + res, stride 1, same padding."""
    _dev(x, "x")
    N, D, H, W, ci = x.shape
    y = torch.empty((N, D, H, W, c_out), dtype=x.dtype, device=x.device)
    a = nat.Conv3dArgs()
    a.x, a.w_packed, a.y = x.data_ptr(), w_packed.data_ptr(), y.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.ab = ab.data_ptr() if ab is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.N, a.D, a.H, a.W, a.C_in, a.C_out = N, D, H, W, ci, c_out
    a.kd, a.kh, a.kw = (int(v) for v in kernel)
    a.act_in, a.act_param = int(act_in), float(act_param)
    a.res_mode = nat.RES_ADD if res is not None else nat.RES_NONE
    a.dtype = dtype_code(x.dtype)
    _run(f"conv3d_fwd[{ci}->{c_out},k{a.kd}{a.kh}{a.kw}]", _nbytes(x, y), nat.lib().pytc_conv3d_fwd, C.byref(a), _stream(),
         flops=2 * N * D * H * W * ci * c_out * a.kd * a.kh * a.kw)
    return y


def channel_stats(x: torch.Tensor) -> torch.Tensor:
    """x (N, *spatial, C) -> partial statistics (N, slots, 2, C)."""
    _dev(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (N * Cc)
    slots = nat.lib().pytc_channel_stats_slots(rows)
    st = torch.empty((N, slots, 2, Cc), dtype=torch.float32, device=x.device)
    _run(f"channel_stats[C{Cc}]", _nbytes(x), nat.lib().pytc_channel_stats, _p(x), _p(st), N, rows, Cc, dtype_code(x.dtype),
         _stream())
    return st


def pad_channels(c: int, dtype: torch.dtype) -> int:
    """Channel count a dense-conv model's activations are carried with: rows 16-byte aligned (8 bf16 / 4 fp32 channels) so that every
    kernel of the path runs its vector form.  The reference's stock RSUNet widths [18, 36, 48, 64, 80] (arch_profiles.yaml:34-44)
    travel as 24 / 40 / 48 / 64 / 80 channels in bf16, the extra ones all-zero (zero weight rows / columns, affine (0, 0), act(0) = 0).
    Counts up to 4 (network inputs, heads) are served by the thin kernels and stay as they are."""
    q = 8 if dtype == torch.bfloat16 else 4
    return c if (c <= 4 or c % q == 0) else (c + q - 1) // q * q


def norm_finalize_groups(stats: torch.Tensor, count: float, gamma, beta, eps: float, groups: int, cpg: int = 0) -> torch.Tensor:
    """cpg > 0: `groups` groups of `cpg` real channels; the channels of `stats` beyond groups * cpg are alignment padding (affine 0)."""
    N, slots, _, Cc = stats.shape
    ab = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    _run("norm_finalize_groups", _nbytes(stats, ab), nat.lib().pytc_norm_finalize_groups_cpg, _p(stats), slots, float(count),
         _p(gamma), _p(beta), float(eps), int(groups), int(cpg), _p(ab), None, N, Cc, _stream())
    return ab


def maxpool3d(x: torch.Tensor, factor) -> torch.Tensor:
    _dev(x, "x")
    N, D, H, W, Cc = x.shape
    fz, fy, fx = (int(v) for v in factor)
    y = torch.empty((N, D // fz, H // fy, W // fx, Cc), dtype=x.dtype, device=x.device)
    _run("maxpool3d", _nbytes(x, y), nat.lib().pytc_maxpool3d_fwd, _p(x), _p(y), N, D, H, W, Cc, fz, fy, fx,
         dtype_code(x.dtype), _stream())
    return y


def dwconvT3d_generic(x: torch.Tensor, w_taps: torch.Tensor, kernel, stride, pad) -> torch.Tensor:
    """Depthwise transposed conv with per-axis geometry; w_taps fp32 (kd*kh*kw, C)."""
    _dev(x, "x")
    N, D, H, W, Cc = x.shape
    k, s, p = ([int(v) for v in t] for t in (kernel, stride, pad))
    out = [(d - 1) * s[i] - 2 * p[i] + k[i] for i, d in enumerate((D, H, W))]
    y = torch.empty((N, *out, Cc), dtype=x.dtype, device=x.device)
    arr = lambda v: (C.c_int32 * 3)(*v)
    _run("dwconvT3d_generic", _nbytes(x, y), nat.lib().pytc_dwconvT3d_generic_fwd, _p(x), _p(y), _p(w_taps), N, D, H, W,
         Cc, arr(k), arr(s), arr(p), dtype_code(x.dtype), _stream())
    return y


def affine_act(x: torch.Tensor, ab: Optional[torch.Tensor], act: int, param: float = 0.0) -> torch.Tensor:
    """y = act(a*x + b) elementwise on (N, *spatial, C)."""
    _dev(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (N * Cc)
    y = torch.empty_like(x)
    _run("affine_act", _nbytes(x, y), nat.lib().pytc_affine_act, _p(x), _p(y), _p(ab), N, rows, Cc, int(act),
         float(param), dtype_code(x.dtype), _stream())
    return y


# ------------------------------------------------------------------ backward (training step)
def _i3(v):
    return (C.c_int32 * 3)(*[int(a) for a in v])


# ---- dense-conv (RSUNet) training ops ---------------------------------------------------------------------------------
def norm_finalize_groups_mr(stats: torch.Tensor, count: float, gamma, beta, eps: float, groups: int, cpg: int = 0):
    """-> (ab (N,2,C), mean_rstd (N,2,C)); cpg as in norm_finalize_groups"""
    N, slots, _, Cc = stats.shape
    ab = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    mr = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    _run("norm_finalize_groups", _nbytes(stats, ab, mr), nat.lib().pytc_norm_finalize_groups_cpg, _p(stats), slots,
         float(count), _p(gamma), _p(beta), float(eps), int(groups), int(cpg), _p(ab), _p(mr), N, Cc, _stream())
    return ab, mr


def conv3d_wgrad(a: torch.Tensor, dy: torch.Tensor, kernel) -> torch.Tensor:
    """a (N,D,H,W,C_in) = the conv's (activated) input, dy (N,D,H,W,C_out) -> dW fp32 (C_out, C_in, kd, kh, kw)."""
    _dev(a, "a"); _dev(dy, "dy")
    N, D, H, W, ci = a.shape
    co = dy.shape[-1]
    kd, kh, kw = (int(v) for v in kernel)
    taps = kd * kh * kw
    k3 = _i3((kd, kh, kw))
    n_ws = nat.lib().pytc_conv3d_wgrad_ws_elems(N, D, H, W, ci, co, k3, dtype_code(a.dtype))
    ws = torch.empty((n_ws,), dtype=torch.float32, device=a.device)
    dW = torch.empty((taps, co, ci), dtype=torch.float32, device=a.device)
    _run(f"conv3d_wgrad[{ci}->{co},k{kd}{kh}{kw}]", taps * _nbytes(a, dy), nat.lib().pytc_conv3d_wgrad, _p(a), _p(dy), _p(dW),
         _p(ws), N, D, H, W, ci, co, k3, dtype_code(a.dtype), _stream(), flops=2 * N * D * H * W * ci * co * taps)
    return dW.view(kd, kh, kw, co, ci).permute(3, 4, 0, 1, 2).contiguous()


def act_bwd(da: torch.Tensor, x: torch.Tensor, ab: Optional[torch.Tensor], act: int, param: float = 0.0, *,
            want_prelu: bool = False):
    """dt = da * act'(a*x + b); with want_prelu also dp = da * min(t, 0) (summed by the caller)."""
    _dev(da, "da"); _dev(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (N * Cc)
    dt = torch.empty_like(x)
    dp = torch.empty_like(x) if want_prelu else None
    _run("act_bwd", _nbytes(da, x, dt), nat.lib().pytc_act_bwd, _p(da), _p(x), _p(ab), _p(dt), _p(dp), N, rows, Cc, int(act),
         float(param), dtype_code(x.dtype), _stream())
    return dt, dp


def norm_bwd_stats(d: torch.Tensor, x: torch.Tensor, mean_rstd: torch.Tensor) -> torch.Tensor:
    """-> (N, 2, C): (sum d, sum d * xhat) per (sample, channel)"""
    _dev(d, "d"); _dev(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (N * Cc)
    ws = torch.empty((nat.lib().pytc_norm_bwd_ws_elems(N, rows, Cc),), dtype=torch.float32, device=x.device)
    s = torch.empty((N, 2, Cc), dtype=torch.float32, device=x.device)
    _run(f"norm_bwd_stats[C{Cc}]", _nbytes(d, x), nat.lib().pytc_norm_bwd_stats, _p(d), _p(x), _p(mean_rstd), _p(ws), _p(s), N,
         rows, Cc, dtype_code(x.dtype), _stream())
    return s


def pw_wgrad_groupnorm_supported(c: int, c_hid: int, dtype: torch.dtype) -> bool:
    return dtype == torch.bfloat16 and bool(nat.lib().pytc_pw_wgrad_groupnorm_supported(int(c), int(c_hid), dtype_code(dtype)))


def pw_wgrad_groupnorm(t: torch.Tensor, mean_rstd: torch.Tensor, ab: torch.Tensor, dhp: torch.Tensor, w2: torch.Tensor,
                       gamma: Optional[torch.Tensor], *, N: int, rows_per_sample: int, c: int, c_hid: int, count: float,
                       defer: Optional["DeferredReduce"] = None):
    """Weight gradient of a GroupNorm-fed expand conv AND the GroupNorm backward from one pass over (t, dhp)
    (pytc_pw_wgrad_groupnorm).  w2: fp32 (c_hid, c).  -> dW2 (c_hid, c), db2 (c_hid), s (N, 2, c) = (sum dtn, sum dtn * xhat),
    coef (N, 3, c) for pw_conv(res_mode=RES_NORM_BWD, res=t, res_bias=coef) of the data-gradient GEMM.  dW2 / db2 are sums over N
    sample terms: with `defer` they join that object's single reduction launch and hold their values only after defer.flush()."""
    _dev(t, "t"); _dev(dhp, "dhp")
    dev = t.device
    lib = nat.lib()
    sps = lib.pytc_pw_wgrad_groupnorm_sps(N, rows_per_sample, c, c_hid)
    nW = c_hid * c
    ws = torch.empty((int(lib.pytc_pw_wgrad_groupnorm_ws_elems(N, rows_per_sample, c, c_hid)),), dtype=torch.float32, device=dev)
    s = torch.empty((N, 2, c), dtype=torch.float32, device=dev)
    coef = torch.empty((N, 3, c), dtype=torch.float32, device=dev)
    _run(f"pw_wgrad_gn[{c}->{c_hid}]", _nbytes(t, dhp), lib.pytc_pw_wgrad_groupnorm, _p(t), _p(mean_rstd), _p(ab), _p(dhp), _p(w2),
         _p(gamma), float(count), _p(s), _p(coef), _p(ws), N, rows_per_sample, c, c_hid, dtype_code(t.dtype), _stream(),
         symbol="pw_wgrad_mfma_kernel")
    term0 = N * sps * (nW + c_hid)
    dW = torch.empty((c_hid, c), dtype=torch.float32, device=dev)
    db = torch.empty((c_hid,), dtype=torch.float32, device=dev)
    own = defer if defer is not None else DeferredReduce()
    tw = 0 if sps == 1 else term0            # one slot per sample: the samples' terms stay in the partials region (see pytc_pw_wgrad_groupnorm)
    own.add(ws[tw:tw + N * nW], dW, nW, N, keep=ws)
    own.add(ws[term0 + N * nW:term0 + N * (nW + c_hid)], db, c_hid, N)
    if defer is None:
        own.flush()
    return dW, db, s, coef


def norm_bwd_apply_supported(x: torch.Tensor) -> bool:
    return x.dtype in (torch.bfloat16, torch.float32) and x.shape[-1] % (8 if x.dtype == torch.bfloat16 else 4) == 0


def norm_bwd_apply(dtn: torch.Tensor, t: torch.Tensor, mean_rstd: torch.Tensor, gamma: Optional[torch.Tensor], s: torch.Tensor, *,
                   count: float, crop_grid=None) -> torch.Tensor:
    """The apply pass of norm_bwd with the statistics given: s (N, 2, C), or (parts, N, 2, C) to be added over the parts.
    crop_grid = (D, H, W) of the rows: the front faces are dropped, -> (N, D-1, H-1, W-1, C); otherwise -> shaped like dtn."""
    _dev(dtn, "dtn"); _dev(t, "t")
    N, Cc = t.shape[0], t.shape[-1]
    rows = t.numel() // (N * Cc)
    parts = s.shape[0] if s.dim() == 4 else 1
    if crop_grid is not None:
        d, h, w = (int(v) for v in crop_grid)
        dt = torch.empty((N, d - 1, h - 1, w - 1, Cc), dtype=t.dtype, device=t.device)
    else:
        dt = torch.empty_like(dtn)
    _run(f"norm_bwd_apply[C{Cc}]", _nbytes(dtn, t, dt), nat.lib().pytc_norm_bwd_apply, _p(dtn), _p(t), _p(mean_rstd), _p(gamma), _p(s),
         parts, _p(dt), N, rows, Cc, float(count), dtype_code(t.dtype), _i3(crop_grid) if crop_grid is not None else None, _stream())
    return dt


def act_norm_bwd_supported(x: torch.Tensor) -> bool:
    return x.shape[-1] % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.dtype in (torch.bfloat16, torch.float32)


def act_norm_bwd_stats(da: torch.Tensor, x: torch.Tensor, ab: Optional[torch.Tensor], mean_rstd: torch.Tensor, act: int, prm: float,
                       want_prelu: bool = False):
    """-> s (N, 2, C) = (sum dt, sum dt * xhat) with dt = da * act'(a*x + b) never stored, p (N, C) | None = sum da * min(t, 0)"""
    _dev(da, "da"); _dev(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (N * Cc)
    n_ws = nat.lib().pytc_norm_bwd_ws_elems(N, rows, Cc)
    ws = torch.empty((n_ws + (n_ws // 2 if want_prelu else 0),), dtype=torch.float32, device=x.device)
    s = torch.empty((N, 2, Cc), dtype=torch.float32, device=x.device)
    p = torch.empty((N, Cc), dtype=torch.float32, device=x.device) if want_prelu else None
    _run(f"act_norm_bwd_stats[C{Cc}]", _nbytes(da, x), nat.lib().pytc_act_norm_bwd_stats, _p(da), _p(x), _p(ab), _p(mean_rstd), _p(ws),
         _p(s), _p(ws[n_ws:]) if want_prelu else None, _p(p), N, rows, Cc, int(act), float(prm), dtype_code(x.dtype), _stream())
    return s, p


def act_norm_bwd_apply(da: torch.Tensor, x: torch.Tensor, ab: Optional[torch.Tensor], mean_rstd: torch.Tensor, gamma, M: torch.Tensor,
                       act: int, prm: float) -> torch.Tensor:
    _dev(da, "da"); _dev(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (N * Cc)
    dx = torch.empty_like(x)
    cg = Cc if gamma is None else int(gamma.numel())      # fewer entries than channels: the rest of x is alignment padding (gamma = 0 there)
    _run(f"act_norm_bwd_apply[C{Cc}]", _nbytes(da, x, dx), nat.lib().pytc_act_norm_bwd_apply_cg, _p(da), _p(x), _p(ab), _p(mean_rstd),
         _p(gamma), cg, _p(M), _p(dx), N, rows, Cc, int(act), float(prm), dtype_code(x.dtype), _stream())
    return dx


def norm_bwd_apply_general(d: torch.Tensor, x: torch.Tensor, mean_rstd: torch.Tensor, gamma, M: torch.Tensor) -> torch.Tensor:
    _dev(d, "d"); _dev(x, "x")
    N, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (N * Cc)
    dx = torch.empty_like(x)
    _run(f"norm_bwd_apply[C{Cc}]", _nbytes(d, x, dx), nat.lib().pytc_norm_bwd_apply_general, _p(d), _p(x), _p(mean_rstd),
         _p(gamma), _p(M), _p(dx), N, rows, Cc, dtype_code(x.dtype), _stream())
    return dx


def maxpool3d_bwd(x: torch.Tensor, dy: torch.Tensor, factor) -> torch.Tensor:
    _dev(x, "x"); _dev(dy, "dy")
    N, D, H, W, Cc = x.shape
    fz, fy, fx = (int(v) for v in factor)
    dx = torch.empty_like(x)
    _run("maxpool3d_bwd", _nbytes(x, dy, dx), nat.lib().pytc_maxpool3d_bwd, _p(x), _p(dy), _p(dx), N, D, H, W, Cc, fz, fy, fx,
         dtype_code(x.dtype), _stream())
    return dx


def dwconv3d_generic(x: torch.Tensor, w_taps: torch.Tensor, kernel, stride, pad, out_dims) -> torch.Tensor:
    """Depthwise conv with per-axis kernel / stride / pad (gather form); w_taps fp32 (kd*kh*kw, C)."""
    _dev(x, "x")
    N, D, H, W, Cc = x.shape
    od = tuple(int(v) for v in out_dims)
    y = torch.empty((N,) + od + (Cc,), dtype=x.dtype, device=x.device)
    _run("dwconv3d_generic", _nbytes(x, y), nat.lib().pytc_dwconv3d_generic_fwd, _p(x), _p(y), _p(w_taps), N, D, H, W, Cc,
         _i3(kernel), _i3(stride), _i3(pad), _i3(od), dtype_code(x.dtype), _stream())
    return y


def groupnorm_finalize_mr(stats: torch.Tensor, count: float, gamma, beta, eps: float = 1e-5):
    """-> (ab (N,2,C), mean_rstd (N,2,C))"""
    N, slots, _, Cc = stats.shape
    ab = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    mr = torch.empty((N, 2, Cc), dtype=torch.float32, device=stats.device)
    _run("groupnorm_finalize", _nbytes(stats, ab), nat.lib().pytc_groupnorm_finalize_mr, _p(stats), slots, float(count),
         _p(gamma), _p(beta), float(eps), _p(ab), _p(mr), N, Cc, _stream())
    return ab, mr


def gelu(x: torch.Tensor, dy: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy is None: gelu(x); else dy * gelu'(x)."""
    _dev(x, "x")
    out = torch.empty_like(x)
    _run(("gelu_bwd" if dy is not None else "gelu_fwd") + f"[C{x.shape[-1]}]", _nbytes(x, out, dy), nat.lib().pytc_gelu, _p(x), _p(dy), _p(out),
         x.numel(), dtype_code(x.dtype), _stream())
    return out


def add_(y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _dev(y, "y"); _dev(x, "x")
    _run(f"add_inplace[C{x.shape[-1]}]", 3 * _nbytes(x), nat.lib().pytc_add_inplace, _p(y), _p(x), y.numel(), dtype_code(y.dtype), _stream())
    return y


def copy_zero_front(x: torch.Tensor) -> torch.Tensor:
    """Copy of x (N, D, H, W, C) with the front faces (z, y or x == 0) zeroed -- one launch (pytc_copy_zero_front)."""
    _dev(x, "x")
    if x.dim() != 5 or not x.is_contiguous() or (x.shape[-1] * x.element_size()) % 16:
        y = x.clone()
        y[:, 0] = 0
        y[:, :, 0] = 0
        y[:, :, :, 0] = 0
        return y
    y = torch.empty_like(x)
    _run(f"copy_zero_front[C{x.shape[-1]}]", 2 * _nbytes(x), nat.lib().pytc_copy_zero_front, _p(x), _p(y), int(x.shape[0]), _i3(x.shape[1:4]),
         int(x.shape[-1]), dtype_code(x.dtype), _stream())
    return y


_ST_CODES = {"uint8": (0, torch.uint8), "int8": (1, torch.int8), "uint16": (2, torch.uint16), "int16": (3, torch.int16),
             "int32": (4, torch.int32), "float16": (5, torch.float16), "float32": (6, torch.float32)}


def scale_cast(x: torch.Tensor, *, scale: float = 1.0, target: str = "float32") -> torch.Tensor:
    """cast(clip(x * scale)) with numpy's clip-then-truncate semantics (see pytc_scale_cast); x fp32, any shape."""
    _dev(x, "x")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("scale_cast expects a contiguous float32 tensor")
    if target not in _ST_CODES:
        raise ValueError(f"scale_cast: unsupported target dtype {target!r}; supported: {sorted(_ST_CODES)}")
    code, tdt = _ST_CODES[target]
    y = torch.empty(x.shape, dtype=tdt, device=x.device)
    _run("scale_cast", _nbytes(x, y), nat.lib().pytc_scale_cast, _p(x), _p(y), x.numel(), float(scale), code, _stream())
    return y


def set_tuning(key: str, value: int) -> None:
    """Kernel-variant knob (A/B measurements and parity tests between variants); see pytc_set_tuning."""
    nat.check(nat.lib().pytc_set_tuning(key.encode(), int(value)), "set_tuning")


class DeferredReduce:
    """Collects the slot partials of several weight gradients (and any other [slots][n] -> [n] sums) of one backward
    function and reduces them in ONE launch (pytc_reduce_slots_multi) instead of one launch per gradient.  The outputs are
    valid after flush(); results are bit-identical to the immediate reductions."""

    MAX_ITEMS = 12

    def __init__(self):
        self.items = []       # (partials view, out, n, slots, out_t)
        self.keep = []        # workspaces the partial views live in

    def add(self, part: torch.Tensor, out: torch.Tensor, n: int, slots: int, keep=None, out_t: int = 0) -> None:
        """out_t > 0: the n sums form a [n / out_t][out_t] matrix and `out` receives its transpose."""
        self.items.append((part, out, int(n), int(slots), int(out_t)))
        if keep is not None:
            self.keep.append(keep)

    def flush(self) -> None:
        # partials / outputs may have been allocated on another stream (training/autograd.py _WgradLane): this launch, and
        # whoever reads the outputs afterwards, use them on the current stream
        cur = torch.cuda.current_stream()
        for part, out, _n, _s, _t in self.items:
            part.record_stream(cur)
            out.record_stream(cur)
        for b0 in range(0, len(self.items), self.MAX_ITEMS):
            chunk = self.items[b0:b0 + self.MAX_ITEMS]
            arr = (nat.ReduceItem * len(chunk))()
            for i, (part, out, n, slots, out_t) in enumerate(chunk):
                arr[i].part, arr[i].out, arr[i].n, arr[i].slots, arr[i].out_t = part.data_ptr(), out.data_ptr(), n, slots, out_t
            _run("reduce_slots_multi", sum(4 * n * (s + 1) for _p0, _o, n, s, _t in chunk), nat.lib().pytc_reduce_slots_multi, arr,
                 len(chunk), _stream())
        self.items, self.keep = [], []


def pw_wgrad(x: torch.Tensor, dy: torch.Tensor, *, N: int, rows_per_sample: int, c_in: int, c_out: int,
             ab: Optional[torch.Tensor] = None, want_bias: bool = True, x_act: int = nat.ACT_NONE,
             defer: Optional[DeferredReduce] = None, in_major: bool = False):
    """-> dW (c_out, c_in) fp32, db (c_out) fp32 | None;  x_act=ACT_GELU: the GEMM operand is gelu(x).  With `defer` the
    slot reduction joins that object's single launch: dW / db hold their values only after defer.flush().
    in_major: dW arrives as (c_in, c_out) -- a ConvTranspose weight's layout (with `defer` the reduction launch writes it transposed)."""
    _dev(x, "x"); _dev(dy, "dy")
    slots = nat.lib().pytc_pw_wgrad_slots(N * rows_per_sample)
    nW = c_out * c_in
    ws = torch.empty((slots * (nW + c_out),), dtype=torch.float32, device=x.device)
    dW = torch.empty((c_in, c_out) if (in_major and defer is not None) else (c_out, c_in), dtype=torch.float32, device=x.device)
    db = torch.empty((c_out,), dtype=torch.float32, device=x.device) if want_bias else None
    if defer is not None:
        used = C.c_int(0)
        _run(f"pw_wgrad[{c_in}->{c_out}]", _nbytes(x, dy), nat.lib().pytc_pw_wgrad_partial, _p(x), _p(ab), _p(dy), _p(ws),
             int(want_bias), N, rows_per_sample, c_in, c_out, dtype_code(x.dtype), int(x_act), C.byref(used), _stream())
        u = int(used.value)
        defer.add(ws[:u * nW], dW, nW, u, keep=ws, out_t=c_in if in_major else 0)
        if want_bias:
            defer.add(ws[u * nW:u * (nW + c_out)], db, c_out, u)
        return dW, db
    _run(f"pw_wgrad[{c_in}->{c_out}]", _nbytes(x, dy), nat.lib().pytc_pw_wgrad, _p(x), _p(ab), _p(dy), _p(dW), _p(db),
         _p(ws), N, rows_per_sample, c_in, c_out, dtype_code(x.dtype), int(x_act), _stream())
    return (dW.t().contiguous() if in_major else dW), db


def pw_wgrad_dgrad_supported(c_in: int, c_out: int, dtype: torch.dtype) -> bool:
    try:
        return bool(nat.lib().pytc_pw_wgrad_dgrad_supported(int(c_in), int(c_out), dtype_code(dtype)))
    except Exception:
        return False


def pw_wgrad_dgrad(hp: torch.Tensor, dy: torch.Tensor, w_t_paired: torch.Tensor, *, N: int, rows_per_sample: int, c_in: int, c_out: int,
                   want_bias: bool = True, defer: Optional[DeferredReduce] = None):
    """One pass over the hidden pre-activation hp (N, rows, c_in) and the output gradient dy (N, rows, c_out) of a mixer's projecting
    conv: -> (dW (c_out, c_in), db (c_out) | None, dhp (N, rows, c_in) bf16) with dW = sum_r dy^T gelu(hp) and
    dhp = (W^T dy) * gelu'(hp); w_t_paired = packed_paired(W, transposed=True).  dW / db bit-identical to pw_wgrad(x_act=GELU); dhp equal to the
    RES_GELU_BWD data-gradient GEMM on that paired image up to one bf16 ulp in a few outputs per million; `defer` as in pw_wgrad."""
    _dev(hp, "hp"); _dev(dy, "dy"); _dev(w_t_paired, "w_t_paired")
    if hp.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16 or w_t_paired.dtype != torch.bfloat16:
        raise TypeError("pw_wgrad_dgrad runs on bfloat16 operands and the bf16 paired image of W^T")
    slots = nat.lib().pytc_pw_wgrad_slots(N * rows_per_sample)
    nW = c_out * c_in
    ws = torch.empty((slots * (nW + c_out),), dtype=torch.float32, device=hp.device)
    dW = torch.empty((c_out, c_in), dtype=torch.float32, device=hp.device)
    db = torch.empty((c_out,), dtype=torch.float32, device=hp.device) if want_bias else None
    dhp = torch.empty((N, rows_per_sample, c_in), dtype=torch.bfloat16, device=hp.device)
    own = defer is None
    if own:
        defer = DeferredReduce()
    used = C.c_int(0)
    _run(f"pw_wgrad_dgrad[{c_in}->{c_out}]", _nbytes(hp, dy, dhp), nat.lib().pytc_pw_wgrad_dgrad_partial, _p(hp), _p(dy), _p(w_t_paired),
         _p(dhp), _p(ws), int(want_bias), N, rows_per_sample, c_in, c_out, dtype_code(hp.dtype), C.byref(used), _stream())
    u = int(used.value)
    defer.add(ws[:u * nW], dW, nW, u, keep=ws)
    if want_bias:
        defer.add(ws[u * nW:u * (nW + c_out)], db, c_out, u)
    if own:
        defer.flush()
    return dW, db, dhp


def mixer_bwd_rc_supported(c: int, c_hid: int, c_out: int, dtype: torch.dtype) -> int:
    """0: no; 1: the form without the GroupNorm sums; 2: both (pytc_mixer_bwd_rc_supported)."""
    try:
        return int(nat.lib().pytc_mixer_bwd_rc_supported(int(c), int(c_hid), int(c_out), dtype_code(dtype)))
    except Exception:
        return 0


def mixer_bwd_rc(t: torch.Tensor, ab: torch.Tensor, dy: torch.Tensor, w2_paired: torch.Tensor, b2: torch.Tensor, w3t_paired: torch.Tensor,
                 *, N: int, rows_per_sample: int, c: int, c_hid: int, c_out: int, want_bias: bool = True,
                 mean_rstd: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
                 count: float = 0.0, defer: Optional["DeferredReduce"] = None):
    """Backward of a full-resolution block's mixer with the hidden pre-activation rebuilt from the depthwise output t (N, rows, c)
    (pytc_mixer_bwd_rc): -> (dW3 (c_out, c_hid), db3 (c_out) | None, dhp (N, rows, c_hid) bf16) and, when mean_rstd / w2 (fp32 (c_hid, c))
    are given (the GroupNorm form), also (dW2 (c_hid, c), db2 (c_hid), s (N, 2, c), coef (N, 3, c)) as pw_wgrad_groupnorm returns them.
    `defer` as in pw_wgrad: the slot sums join that object's reduction launch."""
    _dev(t, "t"); _dev(dy, "dy"); _dev(w2_paired, "w2_paired"); _dev(w3t_paired, "w3t_paired")
    if t.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16 or w2_paired.dtype != torch.bfloat16 or w3t_paired.dtype != torch.bfloat16:
        raise TypeError("mixer_bwd_rc runs on bfloat16 operands and bf16 paired weight images")
    gn = mean_rstd is not None
    if gn and (w2 is None or count <= 0):
        raise ValueError("mixer_bwd_rc: the GroupNorm form takes mean_rstd, w2 (fp32) and the voxel count of the statistics")
    lib, dev = nat.lib(), t.device
    sps = lib.pytc_mixer_bwd_rc_sps(N, rows_per_sample, c_hid)
    S, nW3, nW2 = N * sps, c_out * c_hid, c_hid * c
    ws = torch.empty((int(lib.pytc_mixer_bwd_rc_ws_elems(N, rows_per_sample, c_hid, int(gn))),), dtype=torch.float32, device=dev)
    dhp = torch.empty((N, rows_per_sample, c_hid), dtype=torch.bfloat16, device=dev)
    s = torch.empty((N, 2, c), dtype=torch.float32, device=dev) if gn else None
    coef = torch.empty((N, 3, c), dtype=torch.float32, device=dev) if gn else None
    used = C.c_int(0)
    _run(f"mixer_bwd_rc[{c}->{c_hid}->{c_out}]" + ("+gn" if gn else ""), _nbytes(t, dy, dhp), lib.pytc_mixer_bwd_rc, _p(t), _p(ab),
         _p(mean_rstd), _p(dy), _p(w2_paired), _p(b2), _p(w3t_paired), _p(w2), _p(gamma), float(count), _p(dhp), _p(ws), _p(s), _p(coef),
         int(want_bias), int(gn), N, rows_per_sample, c, c_hid, c_out, dtype_code(t.dtype), C.byref(used), _stream(),
         symbol="mixer_bwd_rc_kernel")
    assert int(used.value) == S
    dW3 = torch.empty((c_out, c_hid), dtype=torch.float32, device=dev)
    db3 = torch.empty((c_out,), dtype=torch.float32, device=dev) if want_bias else None
    own = defer if defer is not None else DeferredReduce()
    own.add(ws[:S * nW3], dW3, nW3, S, keep=ws)
    if want_bias:
        own.add(ws[S * nW3:S * (nW3 + c_out)], db3, c_out, S)
    out = (dW3, db3, dhp)
    if gn:
        term0 = S * (nW3 + c_out) + S * (nW2 + c_hid)
        dW2 = torch.empty((c_hid, c), dtype=torch.float32, device=dev)
        db2 = torch.empty((c_hid,), dtype=torch.float32, device=dev)
        own.add(ws[term0:term0 + N * nW2], dW2, nW2, N)
        own.add(ws[term0 + N * nW2:term0 + N * (nW2 + c_hid)], db2, c_hid, N)
        out = out + (dW2, db2, s, coef)
    if defer is None:
        own.flush()
    return out


def dw_wgrad(g: torch.Tensor, x: torch.Tensor, *, K: int, stride: int = 1, want_bias: bool = True,
             defer: Optional[DeferredReduce] = None, channel_major: bool = False):
    """g (N,*gdims,C), x (N,*xdims,C) -> dW (K^3, C) fp32, db (C) | None   (see pytc_dw_wgrad); `defer` as in pw_wgrad;
    channel_major: dW arrives as (C, K^3), the parameter's own layout (with `defer` the reduction launch writes it transposed)"""
    _dev(g, "g"); _dev(x, "x")
    N, Cc = g.shape[0], g.shape[-1]
    gd, xd = _i3(g.shape[1:4]), _i3(x.shape[1:4])
    slots = nat.lib().pytc_dw_wgrad_slots(N, gd, xd, Cc, K, stride, dtype_code(g.dtype))
    if slots < 0:
        raise RuntimeError(f"dw_wgrad: unsupported channel count {Cc}")
    nW = K ** 3 * Cc
    ws = torch.empty((slots * (nW + Cc),), dtype=torch.float32, device=g.device)
    dW = torch.empty((Cc, K ** 3) if (channel_major and defer is not None) else (K ** 3, Cc), dtype=torch.float32, device=g.device)
    db = torch.empty((Cc,), dtype=torch.float32, device=g.device) if want_bias else None
    if defer is not None:
        used = C.c_int(0)
        _run(f"dw_wgrad[C{Cc}_k{K}]", _nbytes(g, x), nat.lib().pytc_dw_wgrad_partial, _p(g), _p(x), _p(ws), int(want_bias), N,
             gd, xd, Cc, K, stride, dtype_code(g.dtype), C.byref(used), _stream())
        u = int(used.value)
        defer.add(ws[:u * nW], dW, nW, u, keep=ws, out_t=Cc if channel_major else 0)
        if want_bias:
            defer.add(ws[u * nW:u * (nW + Cc)], db, Cc, u)
        return dW, db
    _run(f"dw_wgrad[C{Cc}_k{K}]", _nbytes(g, x), nat.lib().pytc_dw_wgrad, _p(g), _p(x), _p(dW), _p(db), _p(ws), N, gd, xd,
         Cc, K, stride, dtype_code(g.dtype), _stream())
    return (dW.t().contiguous() if channel_major else dW), db


def norm_bwd(dtn: torch.Tensor, t: torch.Tensor, mean_rstd: torch.Tensor, gamma: Optional[torch.Tensor],
             count: Optional[float] = None):
    """-> dt (like t), s (N,2,C) with (sum dtn, sum dtn*xhat); count = voxels the forward statistics covered"""
    _dev(dtn, "dtn"); _dev(t, "t")
    N, Cc = t.shape[0], t.shape[-1]
    rows = t.numel() // (N * Cc)
    ws = torch.empty((nat.lib().pytc_norm_bwd_ws_elems(N, rows, Cc),), dtype=torch.float32, device=t.device)
    s = torch.empty((N, 2, Cc), dtype=torch.float32, device=t.device)
    dt = torch.empty_like(t)
    _run(f"norm_bwd[C{t.shape[-1]}]", 3 * _nbytes(t) + _nbytes(dtn), nat.lib().pytc_norm_bwd, _p(dtn), _p(t), _p(mean_rstd), _p(gamma),
         _p(ws), _p(s), _p(dt), N, rows, float(count if count is not None else rows), Cc, dtype_code(t.dtype), _stream())
    return dt, s


def dwconv3d_bwd_data(dy: torch.Tensor, w_taps: torch.Tensor, xdims, *, K: int, stride: int, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Data gradient of a (strided) depthwise conv; add (shaped like the result): dx = conv^T(dy) + add in the same launch."""
    _dev(dy, "dy")
    N, Cc = dy.shape[0], dy.shape[-1]
    dx = torch.empty((N, *[int(v) for v in xdims], Cc), dtype=dy.dtype, device=dy.device)
    if add is not None:
        if tuple(add.shape) != tuple(dx.shape) or add.dtype != dx.dtype or not add.is_contiguous() or (Cc * dx.element_size()) % 16:
            raise ValueError("dwconv3d_bwd_data: `add` must be a contiguous tensor shaped like the result with 16-byte channel groups")
        _run(f"dwconv3d_bwd_data[C{Cc}_k{K}_s{stride}]+add", _nbytes(dy, dx, add), nat.lib().pytc_dwconv3d_bwd_data_add, _p(dy), _p(w_taps),
             _p(add), _p(dx), N, _i3(xdims), _i3(dy.shape[1:4]), Cc, K, stride, dtype_code(dy.dtype), _stream())
        return dx
    _run(f"dwconv3d_bwd_data[C{Cc}_k{K}_s{stride}]", _nbytes(dy, dx), nat.lib().pytc_dwconv3d_bwd_data, _p(dy), _p(w_taps),
         _p(dx), N, _i3(xdims), _i3(dy.shape[1:4]), Cc, K, stride, dtype_code(dy.dtype), _stream())
    return dx


# ---- strided / transposed dense conv (MONAI-style residual U-Net) ------------------------------------------------------
def conv3d_pack_weight_direct(w: torch.Tensor, dtype: torch.dtype, *, layout: str) -> torch.Tensor:
    """fp32 5-D weight -> packed image for conv3d_strided.  layout:
      'conv'      w is nn.Conv3d [C_out][C_in][k]: forward of a (strided) conv
      'convT'     w is nn.ConvTranspose3d [C_in][C_out][k]: forward of the transposed conv (gather form)
      'conv_dgrad'  w is nn.Conv3d [C_out][C_in][k]: data gradient of that conv (transposed gather, C_out -> C_in)
      'convT_dgrad' w is nn.ConvTranspose3d [C_in][C_out][k]: data gradient of that transposed conv (strided conv, C_out -> C_in)"""
    _dev(w, "w")
    a, b, kd, kh, kw = w.shape
    ntap = kd * kh * kw
    if layout == "conv":
        co, ci, s_o, s_c = a, b, b * ntap, ntap
    elif layout == "convT":
        co, ci, s_o, s_c = b, a, ntap, b * ntap
    elif layout == "conv_dgrad":          # out channels of the gradient conv = C_in (b), its in channels = C_out (a)
        co, ci, s_o, s_c = b, a, ntap, b * ntap
    elif layout == "convT_dgrad":         # out channels = C_in_T (a), in channels = C_out_T (b)
        co, ci, s_o, s_c = a, b, b * ntap, ntap
    else:
        raise ValueError(f"unknown weight layout {layout!r}")
    n = nat.lib().pytc_conv3d_direct_packed_elems(co, ci, kd, kh, kw, dtype_code(dtype))
    packed = torch.empty((n,), dtype=dtype, device=w.device)
    _run("conv3d_pack_weight_direct", _nbytes(w, packed), nat.lib().pytc_conv3d_pack_weight_direct, _p(w), co, ci, kd, kh, kw,
         s_o, s_c, 0, _p(packed), dtype_code(dtype), _stream())
    return packed


def conv3d_strided(x: torch.Tensor, w_packed: torch.Tensor, *, c_out: int, kernel, stride, pad, out_dims, transposed: bool,
                   bias: Optional[torch.Tensor] = None, ab: Optional[torch.Tensor] = None, act_in: int = nat.ACT_NONE,
                   act_param: float = 0.0, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (N,Di,Hi,Wi,C_in) -> (N,*out_dims,C_out): strided conv (transposed=False) or ConvTranspose3d in gather form."""
    _dev(x, "x")
    N, Di, Hi, Wi, ci = x.shape
    Do, Ho, Wo = (int(v) for v in out_dims)
    y = torch.empty((N, Do, Ho, Wo, c_out), dtype=x.dtype, device=x.device)
    a = nat.Conv3dArgs()
    a.x, a.w_packed, a.y = x.data_ptr(), w_packed.data_ptr(), y.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.ab = ab.data_ptr() if ab is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.N, a.D, a.H, a.W, a.C_in, a.C_out = N, Do, Ho, Wo, ci, c_out
    a.kd, a.kh, a.kw = (int(v) for v in kernel)
    a.act_in, a.act_param = int(act_in), float(act_param)
    a.res_mode = nat.RES_ADD if res is not None else nat.RES_NONE
    a.dtype = dtype_code(x.dtype)
    tag = "convT3d" if transposed else "conv3d_s"
    _run(f"{tag}_fwd[{ci}->{c_out},k{a.kd}{a.kh}{a.kw}]", _nbytes(x, y), nat.lib().pytc_conv3d_strided_fwd, C.byref(a),
         _i3((Di, Hi, Wi)), _i3(stride), _i3(pad), 1 if transposed else 0, _stream(),
         # useful MACs: a transposed stride-2 conv reaches an output through (3/2)^3 taps on average, a strided one through all
         flops=int(2 * N * Do * Ho * Wo * ci * c_out * a.kd * a.kh * a.kw * (0.125 if transposed else 1.0)))
    return y


def convT3d_phase_supported(c_out: int, c_in: int, dtype: torch.dtype) -> bool:
    """the LDS-tiled phase form covers this stride-2 transposed gather (bf16, C_in % 8 == 0, knob convT_phase_tile)"""
    return dtype == torch.bfloat16 and bool(nat.lib().pytc_convT3d_phase_supported(int(c_out), int(c_in), nat.BF16))


def convT3d_phase(x: torch.Tensor, w_images: torch.Tensor, *, c_out: int, bias: Optional[torch.Tensor] = None,
                  ab: Optional[torch.Tensor] = None, act_in: int = nat.ACT_NONE, act_param: float = 0.0,
                  res: Optional[torch.Tensor] = None, tag: str = "convT3d") -> torch.Tensor:
    """x (N,D,H,W,C_in) bf16 -> (N,2D,2H,2W,C_out): the k 3 / stride 2 / pad 1 transposed gather as eight stride-1 convs on the LDS-tiled
    kernel (pytc_convT3d_phase_fwd); w_images from the 'convT_phase' / 'conv_dgrad_phase' layouts."""
    _dev(x, "x"); _dev(w_images, "w_images")
    N, Di, Hi, Wi, ci = x.shape
    y = torch.empty((N, 2 * Di, 2 * Hi, 2 * Wi, c_out), dtype=x.dtype, device=x.device)
    a = nat.Conv3dArgs()
    a.x, a.w_packed, a.y = x.data_ptr(), w_images.data_ptr(), y.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.ab = ab.data_ptr() if ab is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.N, a.D, a.H, a.W, a.C_in, a.C_out = N, 2 * Di, 2 * Hi, 2 * Wi, ci, c_out
    a.kd, a.kh, a.kw = 3, 3, 3
    a.act_in, a.act_param = int(act_in), float(act_param)
    a.res_mode = nat.RES_ADD if res is not None else nat.RES_NONE
    a.dtype = dtype_code(x.dtype)
    _run(f"{tag}_fwd[{ci}->{c_out},k333]", _nbytes(x, y), nat.lib().pytc_convT3d_phase_fwd, C.byref(a), _i3((Di, Hi, Wi)), _stream(),
         flops=int(2 * N * 8 * Di * Hi * Wi * ci * c_out * 27 * 0.125))
    return y


def convT3d_thin_supported(c_in: int, c_out: int) -> bool:
    return bool(nat.lib().pytc_convT3d_thin_supported(int(c_in), int(c_out)))


def convT3d_thin(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """ConvTranspose3d(k 3, s 2, p 1, op 1) for C_out <= 4: x (N,D,H,W,C_in) -> (N,2D,2H,2W,C_out); w fp32 (C_in, C_out, 3,3,3)."""
    _dev(x, "x"); _dev(w, "w")
    N, D, H, W, ci = x.shape
    co = w.shape[1]
    if tuple(w.shape) != (ci, co, 3, 3, 3) or w.dtype != torch.float32 or not w.is_contiguous():
        raise ValueError(f"convT3d_thin: weight must be contiguous fp32 (C_in={ci}, C_out, 3, 3, 3), got {tuple(w.shape)} {w.dtype}")
    y = torch.empty((N, 2 * D, 2 * H, 2 * W, co), dtype=x.dtype, device=x.device)
    if co == 1 and ci % 8 == 0 and ci <= 512:
        # input-centric: 27 tap products per input voxel, then a 1 .. 8-term gather per output voxel (pytc_convT3d_c1_fwd)
        ws = torch.empty((27 * N * D * H * W,), dtype=torch.float32, device=x.device)
        _run(f"convT3d_fwd[{ci}->{co},k333]", _nbytes(x, y) + 2 * _nbytes(ws), nat.lib().pytc_convT3d_c1_fwd, _p(x), _p(w), _p(bias), _p(y), _p(ws),
             N, _i3((D, H, W)), ci, dtype_code(x.dtype), _stream(), flops=int(2 * N * D * H * W * ci * 27))
        return y
    _run(f"convT3d_fwd[{ci}->{co},k333]", _nbytes(x, y), nat.lib().pytc_convT3d_thin_fwd, _p(x), _p(w), _p(bias), _p(y), N,
         _i3((D, H, W)), ci, co, dtype_code(x.dtype), _stream())
    return y


def conv3d_wgrad_strided(big: torch.Tensor, small: torch.Tensor, kernel, stride, pad) -> torch.Tensor:
    """dW fp32 (C_small, C_big, kd, kh, kw) = sum_r small[r][o] * big[r*stride + tap - pad][k] (see pytc_hip.h)."""
    _dev(big, "big"); _dev(small, "small")
    N = big.shape[0]
    ck, co = big.shape[-1], small.shape[-1]
    kd, kh, kw = (int(v) for v in kernel)
    taps = kd * kh * kw
    n_ws = nat.lib().pytc_conv3d_wgrad_strided_ws_elems(N, _i3(small.shape[1:4]), ck, co, _i3((kd, kh, kw)))
    ws = torch.empty((n_ws,), dtype=torch.float32, device=big.device)
    dW = torch.empty((taps, co, ck), dtype=torch.float32, device=big.device)
    _run(f"conv3d_wgrad_strided[{ck}x{co},k{kd}{kh}{kw}]", taps * _nbytes(big, small), nat.lib().pytc_conv3d_wgrad_strided,
         _p(big), _p(small), _p(dW), _p(ws), N, _i3(big.shape[1:4]), _i3(small.shape[1:4]), ck, co, _i3((kd, kh, kw)),
         _i3(stride), _i3(pad), dtype_code(big.dtype), _stream(),
         flops=2 * N * int(small.shape[1]) * int(small.shape[2]) * int(small.shape[3]) * ck * co * taps)
    return dW.view(kd, kh, kw, co, ck).permute(3, 4, 0, 1, 2).contiguous()
