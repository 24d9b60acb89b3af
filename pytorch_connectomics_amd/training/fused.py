"""Train-step epilogue on HIP kernels (csrc/loss_optim_kernels.hip):

* `bce_dice_loss` — WeightedBCEWithLogitsLoss (models/losses/losses.py:190-266, reduction='mean') + MONAI
  DiceLoss(sigmoid=True) (profiles/loss_profiles.yaml:2-9) as ONE reduction pass forward and ONE elementwise pass
  backward, on the network's channels-last output viewed as NCDHW (strided operands, no layout copy).
* `FusedAdamW` — torch.optim.AdamW semantics (training/optimization/build.py:86-130) with the global-norm gradient clip
  (trainer.py:321 `gradient_clip_val`) and an optional EMA of the parameters (callbacks.py:869-907) folded into one
  multi-tensor kernel over a device pointer table; the clip coefficient never leaves the device.  State layout
  (`step`, `exp_avg`, `exp_avg_sq`) equals torch.optim.AdamW's, so optimizer state dicts load either way.
No CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from .. import _native as nat


def _stream():
    from ..hip_ops import _stream as raw_stream          # the raw-handle getter (no Stream object per launch)
    return raw_stream()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _ncr_strides(t: torch.Tensor):
    """(stride_n, stride_c, stride_r) of a (N, C, *spatial) tensor whose spatial dims collapse into one stride, else None."""
    st, sh = t.stride(), t.shape
    for i in range(2, t.dim() - 1):
        if sh[i + 1] != 1 and sh[i] != 1 and st[i] != st[i + 1] * sh[i + 1]:
            return None
    return (st[0], st[1], st[-1] if t.dim() > 2 else 1)


def _prep(t: torch.Tensor, like: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(HIP) tensor: pytorch_connectomics_amd has no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape != like.shape:
        t = t.expand_as(like)
    s = _ncr_strides(t)
    if s is None:
        t = t.contiguous()
        s = _ncr_strides(t)
    return t, (C.c_int64 * 3)(*s)


class BceDiceLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, w_bce, w_dice, pos_weight, smooth_nr, smooth_dr):
        x, xs = _prep(logits, logits, "logits")
        t, ts = _prep(target, logits, "target")
        w, ws = (None, None) if weight is None else _prep(weight, logits, "weight")
        N, Cc = x.shape[0], x.shape[1]
        R = x.numel() // (N * Cc)
        lib = nat.lib()
        wsz = lib.pytc_bce_dice_ws_elems(N, Cc, R)
        work = torch.empty((wsz,), dtype=torch.float32, device=x.device)
        sums = torch.empty((N * Cc, 5), dtype=torch.float32, device=x.device)
        out = torch.empty((4,), dtype=torch.float32, device=x.device)
        prm = (float(1.0 if pos_weight is None else pos_weight), float(w_bce), float(w_dice), float(smooth_nr), float(smooth_dr))
        nat.check(lib.pytc_bce_dice_fwd(_p(x), _p(t), _p(w), N, Cc, R, xs, ts, ws, *prm, _p(work), _p(sums), _p(out), _stream()),
                  "bce_dice_fwd")
        ctx.save_for_backward(x, t, w if w is not None else x.new_zeros(0), sums, out)
        ctx.prm = prm
        parts = out.detach()
        ctx.mark_non_differentiable(parts)
        return out[0], parts

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        x, t, w, sums, out = ctx.saved_tensors
        w = w if w.numel() else None
        N, Cc = x.shape[0], x.shape[1]
        R = x.numel() // (N * Cc)
        dx = torch.empty_like(x)                       # keeps the (channels-last) strides of the logits
        ds = _ncr_strides(dx)
        if ds is None:
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            ds = _ncr_strides(dx)
        xs = (C.c_int64 * 3)(*_ncr_strides(x))
        ts = (C.c_int64 * 3)(*_ncr_strides(t))
        ws = None if w is None else (C.c_int64 * 3)(*_ncr_strides(w))
        g = g_loss.reshape(1).float().contiguous()
        nat.check(nat.lib().pytc_bce_dice_bwd(_p(x), _p(t), _p(w), _p(sums), _p(out), _p(g), _p(dx), N, Cc, R, xs, ts, ws,
                                              (C.c_int64 * 3)(*ds), *ctx.prm, _stream()), "bce_dice_bwd")
        return dx, None, None, None, None, None, None, None


def bce_dice_loss(logits: torch.Tensor, target: torch.Tensor, weight: Optional[torch.Tensor] = None, *, w_bce: float = 1.0,
                  w_dice: float = 1.0, pos_weight: Optional[float] = None, smooth_nr: float = 1e-5, smooth_dr: float = 1e-5):
    """-> (loss, parts) with parts = (loss, bce term, dice term, bce denominator) detached.  logits (N, C, *spatial)."""
    return BceDiceLossFn.apply(logits, target, weight, w_bce, w_dice, pos_weight, smooth_nr, smooth_dr)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW update + global-norm clip (+ EMA) in two multi-tensor launches."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: float = 0.0, ema_decay: Optional[float] = None, ema_warmup_steps: int = 0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = float(max_grad_norm or 0.0)
        self.ema_decay = ema_decay
        self.ema_warmup_steps = int(ema_warmup_steps)
        self.ema: Dict[torch.nn.Parameter, torch.Tensor] = {}
        self.ema_updates = 0
        self.last_grad_norm: Optional[torch.Tensor] = None      # device scalar, no sync
        self._key = None
        self._tab = self._chunks = self._work = self._nc = None
        self._gptrs = None
        self._norm_coef = None

    # ---- EMA (callbacks.py:732-935: seeded from the live weights, decay 0 during warm-up, swapped in for evaluation) ----
    def ema_state_dict(self, model: torch.nn.Module) -> Dict[str, torch.Tensor]:
        names = {p: n for n, p in model.named_parameters()}
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for p, e in self.ema.items():
            if p in names:
                sd[names[p]] = e.detach().clone()
        return sd

    _RING = 4

    def _tables(self, items):
        """Pointer table [tensor][p, g, m, v, ema, numel, row] and chunk list in HBM.  The gradients of a step are fresh allocations,
        so the g column changes from step to step: the static columns are built once per parameter set, the table travels through a
        ring of pinned host buffers with a non-blocking copy -- the step never waits for the device (a pageable `torch.tensor(..).to(dev)`
        drained the stream every step: the GPU then idled while the host built the table and walked into the next forward, ~0.9 ms of a
        23.5 ms MedNeXt-S step)."""
        skey = tuple((p.data_ptr(), gi) for p, _g, gi in items) + (len(self.ema),)
        gptrs = [g.data_ptr() for _p, g, _gi in items]
        dev = items[0][0].device
        if skey != self._key:
            chunk = nat.lib().pytc_opt_chunk_elems()
            rows, chunks = [], []
            for ti, (p, g, gi) in enumerate(items):
                st = self.state[p]
                e = self.ema.get(p)
                rows.append([p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             0 if e is None else e.data_ptr(), p.numel(), gi])
                chunks.extend((ti, c) for c in range((p.numel() + chunk - 1) // chunk))
            self._static = torch.tensor(rows, dtype=torch.int64)
            pin = dev.type == "cuda"
            self._ring = [torch.empty_like(self._static).pin_memory() if pin else torch.empty_like(self._static) for _ in range(self._RING)]
            self._ring_ev = [None] * self._RING
            self._ring_i = 0
            self._tab = torch.empty(self._static.shape, dtype=torch.int64, device=dev)
            ch = torch.tensor(chunks, dtype=torch.int32)
            self._chunks_host = ch.pin_memory() if pin else ch
            self._chunks = self._chunks_host.to(dev, non_blocking=True)
            self._nc = len(chunks)
            self._work = torch.empty((self._nc,), dtype=torch.float32, device=dev)
            if self._norm_coef is None or self._norm_coef.device != dev:
                self._norm_coef = torch.empty((2,), dtype=torch.float32, device=dev)
            self._key = skey
            self._gptrs = None
        if gptrs == self._gptrs:
            return
        i = self._ring_i
        self._ring_i = (i + 1) % self._RING
        if self._ring_ev[i] is not None:
            self._ring_ev[i].synchronize()          # the copy out of this buffer, RING steps ago, has long finished
        host = self._ring[i]
        host.copy_(self._static)
        host[:, 1] = torch.tensor(gptrs, dtype=torch.int64)
        self._tab.copy_(host, non_blocking=True)
        if dev.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._ring_ev[i] = ev
        self._gptrs = gptrs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        items, gtab, vrow = [], [], {}
        self.ema_updates += 1
        d = 0.0 if self.ema_decay is None else (0.0 if self.ema_updates <= self.ema_warmup_steps else float(self.ema_decay))
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdamW: parameters must be fp32 CUDA(HIP) tensors (no CPU path)")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdamW does not support sparse gradients")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdamW: parameters must be contiguous")
                g = p.grad if p.grad.is_contiguous() and p.grad.dtype == torch.float32 else None
                if g is None:
                    p.grad = p.grad.float().contiguous()
                    g = p.grad
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if self.ema_decay is not None and p not in self.ema:
                    self.ema[p] = p.detach().clone()
                # steps are counted PER PARAMETER like torch.optim.AdamW (a parameter that starts receiving gradients later --
                # an unused head under DDP -- gets its own bias correction): parameters of a group that share a step count
                # share one row of the scalar table, a different count opens another row
                t = int(st.get("_t", 0)) + 1
                st["_t"] = t
                key = (gi, t)
                if key not in vrow:
                    vrow[key] = len(gtab)
                    gtab.append([group["lr"], b1, b2, group["eps"], group["weight_decay"], 1.0 - b1 ** t,
                                 math.sqrt(1.0 - b2 ** t), d])
                items.append((p, g, vrow[key]))
        if not items:
            return loss
        self._tables(items)
        if len(gtab) > 8:
            raise RuntimeError("FusedAdamW: at most 8 (parameter group, step count) rows per update (merge groups with equal "
                               "hyper-parameters)")
        groups = (C.c_float * (8 * len(gtab)))(*[float(v) for row in gtab for v in row])
        lib = nat.lib()
        nat.check(lib.pytc_grad_norm_multi(_p(self._tab), _p(self._chunks), self._nc, self.max_grad_norm, _p(self._work),
                                           _p(self._norm_coef), _stream()), "grad_norm_multi")
        self.last_grad_norm = self._norm_coef[0]
        nat.check(lib.pytc_adamw_multi(_p(self._tab), _p(self._chunks), self._nc, groups, len(gtab), _p(self._norm_coef),
                                       _stream()), "adamw_multi")
        # the kernel wrote the parameters through raw pointers: tell autograd / every (data_ptr, _version)-keyed cache
        # (the models' repacked-weight caches) that their contents changed
        torch.autograd.graph.increment_version([it[0] for it in items])
        return loss

    def state_dict(self):
        for st in self.state.values():
            if "_t" in st:
                st["step"] = torch.tensor(float(st["_t"]))
        sd = super().state_dict()
        # `_t` is this optimizer's host-side counter, `step` the interchange field.  Optimizer.state_dict() hands out the LIVE
        # per-parameter dicts, so the counter is dropped from copies, never from the optimizer's own state
        sd["state"] = {k: {kk: vv for kk, vv in st.items() if kk != "_t"} for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._key = None
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    if "step" in st:
                        st["_t"] = int(float(st["step"]))
                    for k in ("exp_avg", "exp_avg_sq"):
                        if k in st:
                            st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()

    def load_ema_state_dict(self, model: torch.nn.Module, ema_state: Dict[str, torch.Tensor], updates: int = 0) -> None:
        """Resume the EMA shadow weights (the reference callback's `ema_state` / `updates`, callbacks.py:732-790)."""
        if self.ema_decay is None:
            return
        owned = {id(p) for g in self.param_groups for p in g["params"]}
        for n, p in model.named_parameters():
            if id(p) in owned and n in ema_state:
                self.ema[p] = ema_state[n].detach().to(device=p.device, dtype=torch.float32).contiguous().clone()
        self.ema_updates = int(updates)
        self._key = None


__all__ = ["bce_dice_loss", "BceDiceLossFn", "FusedAdamW"]
