"""MedNeXt trunk for MI355X -- the counterpart of the third-party ``nnunet_mednext`` package the
reference imports (connectomics/models/architectures/mednext_models.py:23-32).

Module / parameter names follow the published MedNeXt v1 layout so that reference checkpoints
(``model.model.stem.weight``, ``...enc_block_0.0.conv1.weight`` ...) load with
``strict=True``.  The nn.Conv3d / nn.GroupNorm / nn.ConvTranspose3d children are PARAMETER HOLDERS
only (they also keep torch.optim's "no weight decay on norm modules" grouping working,
training/optimization/build.py:73-112); ``forward`` never calls them -- it runs the hand-written
gfx950 kernels of libpytc_hip.so on NDHWC activations:

  block      : dwconv3d (+ per-(n,c) sum/sumsq)  ->  groupnorm_finalize  ->
               [norm-apply . 1x1 expand . GELU]  ->  [1x1 project + residual]        (MFMA GEMMs)
  down block : same with stride-2 depthwise conv; residual = strided 1x1 conv (gather GEMM)
  up block   : depthwise transposed conv written at +1 offset (the F.pad((1,0)*3) is free),
               residual = transposed 1x1 conv evaluated at low resolution and gathered in the
               epilogue together with the encoder skip.

There is no CPU path: calling ``forward`` with a CPU tensor raises.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn

from ... import _native as nat
from ... import hip_ops as ops

# a block whose mixer sees fewer voxel rows than this (N * voxels: the 7^3 / 14^3 levels of 8 windows of 112^3) runs expand and
# project as two single GEMMs instead of the fused mixer (see _block).  MEASURED on the whole-volume bench (round 3,
# tools/r03_ab_infer.sh): 16384 rows 7.70 -> 7.81 ms per 8 windows with two window streams, 8.79 -> 8.75 with one; 32768 rows 7.96 /
# 8.91 -- the training step gains from this schedule (training/autograd.py FUSED_TRAIN_MIXER_MIN_ROWS), the inference engine does
# not (no hidden tensor to store, and the second window stream already fills the idle CUs).  0 = off (default).
SMALL_ROWS_TWO_GEMMS = int(os.environ.get("PYTC_MIXER_TWO_GEMM_ROWS", "0"))

# bf16 inference: samples per depth-first chain at the FULL-RESOLUTION level (MedNeXt.features_cl).  The level-0 tensors of an
# 8-window batch (0.72 GB each at 112^3 x 32 channels) are far larger than the 256 MiB Infinity Cache, so every kernel streams its
# operands from HBM; run sample slice by sample slice (stem + encoder blocks + down block, later up block + decoder blocks + head), a
# slice's depthwise output / block output (90 MB per window) could be re-read by the next kernel of the chain while still
# cache-resident.  MEASURED on the whole-volume bench (round 3, profiles/r03_l0_subbatch.txt): no such gain -- 7.57 ms per 8 windows
# for the whole batch against 7.60 / 7.79 / 7.98 for slices of 4 / 2 / 1 windows with two window streams (8.58 against 8.67 / 9.12 /
# 10.0 with one): smaller launches fill the chip worse and nothing comes back from the cache to pay for it.  Results are bit-identical
# for every value (no kernel lets a sample's arithmetic depend on the batch it travels in; tests/test_gpu_mednext.py::
# test_level0_subbatch_is_bit_identical), so the switch stays as a memory knob (a slice's level-0 temporaries instead of the
# batch's).  0 = off (default).
L0_SUBBATCH = int(os.environ.get("PYTC_L0_SUBBATCH", "0"))


def _conv_nd(dim: str):
    if dim == "3d":
        return nn.Conv3d, nn.ConvTranspose3d
    if dim == "2d":
        return nn.Conv2d, nn.ConvTranspose2d
    raise ValueError(f"dim must be '2d' or '3d', got {dim!r}")


class _ChannelLayerNorm(nn.Module):
    """channels_first LayerNorm parameter holder (norm_type='layer')."""

    def __init__(self, normalized_shape: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps


class MedNeXtBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, exp_r: int = 4, kernel_size: int = 7,
                 do_res: bool = True, norm_type: str = "group", n_groups: Optional[int] = None,
                 dim: str = "3d", grn: bool = False):
        super().__init__()
        conv, _ = _conv_nd(dim)
        self.do_res = do_res
        self.dim = dim
        self.grn = grn
        self.conv1 = conv(in_channels, in_channels, kernel_size=kernel_size, stride=1,
                          padding=kernel_size // 2, groups=in_channels if n_groups is None else n_groups)
        if norm_type == "group":
            self.norm = nn.GroupNorm(num_groups=in_channels, num_channels=in_channels)
        elif norm_type == "layer":
            self.norm = _ChannelLayerNorm(in_channels)
        else:
            raise ValueError(f"norm_type must be 'group' or 'layer', got {norm_type!r}")
        self.conv2 = conv(in_channels, exp_r * in_channels, kernel_size=1, stride=1, padding=0)
        self.act = nn.GELU()
        self.conv3 = conv(exp_r * in_channels, out_channels, kernel_size=1, stride=1, padding=0)
        if grn:
            shape = (1, exp_r * in_channels, 1, 1, 1) if dim == "3d" else (1, exp_r * in_channels, 1, 1)
            self.grn_beta = nn.Parameter(torch.zeros(shape))
            self.grn_gamma = nn.Parameter(torch.zeros(shape))
        self.kind = "block"

    def forward(self, x, dummy_tensor=None):  # pragma: no cover - guard only
        raise RuntimeError("MedNeXt blocks execute through MedNeXt.forward (HIP engine); they are parameter "
                           "holders and cannot be called directly")


class MedNeXtDownBlock(MedNeXtBlock):
    def __init__(self, in_channels, out_channels, exp_r=4, kernel_size=7, do_res=False, norm_type="group",
                 dim="3d", grn=False):
        super().__init__(in_channels, out_channels, exp_r, kernel_size, do_res=False, norm_type=norm_type,
                         dim=dim, grn=grn)
        conv, _ = _conv_nd(dim)
        self.resample_do_res = do_res
        if do_res:
            self.res_conv = conv(in_channels, out_channels, kernel_size=1, stride=2)
        self.conv1 = conv(in_channels, in_channels, kernel_size=kernel_size, stride=2,
                          padding=kernel_size // 2, groups=in_channels)
        self.kind = "down"


class MedNeXtUpBlock(MedNeXtBlock):
    def __init__(self, in_channels, out_channels, exp_r=4, kernel_size=7, do_res=False, norm_type="group",
                 dim="3d", grn=False):
        super().__init__(in_channels, out_channels, exp_r, kernel_size, do_res=False, norm_type=norm_type,
                         dim=dim, grn=grn)
        _, convt = _conv_nd(dim)
        self.resample_do_res = do_res
        if do_res:
            self.res_conv = convt(in_channels, out_channels, kernel_size=1, stride=2)
        self.conv1 = convt(in_channels, in_channels, kernel_size=kernel_size, stride=2,
                           padding=kernel_size // 2, groups=in_channels)
        self.kind = "up"


class OutBlock(nn.Module):
    def __init__(self, in_channels: int, n_classes: int, dim: str = "3d"):
        super().__init__()
        _, convt = _conv_nd(dim)
        self.conv_out = convt(in_channels, n_classes, kernel_size=1)

    def forward(self, x, dummy_tensor=None):  # pragma: no cover - guard only
        raise RuntimeError("OutBlock executes through MedNeXt.forward (HIP engine)")


# --------------------------------------------------------------------------------------------
class _WeightCache:
    """Device-side repacked parameters, keyed by (parameter identity, version, compute dtype)."""

    def __init__(self):
        self._store: Dict = {}

    def get(self, key, params: Sequence[torch.Tensor], make):
        sig = tuple((p.data_ptr(), p._version) for p in params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            val = make()
        self._store[key] = (sig, val)
        return val

    def clear(self):
        self._store.clear()


def _f32(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if p is None else p.detach().float().contiguous()


class HipBlockOps:
    """Forward of the three MedNeXt block kinds on NDHWC tensors using libpytc_hip kernels."""

    def __init__(self):
        self.cache = _WeightCache()
        self.fused = True      # bf16: use the fused channel-mixer kernel where a template exists
        # bf16 up blocks: depthwise transposed conv recomputed in the mixer's prologue, its 2C-channel high-resolution
        # output never written (pw_mlp_up_kernels.hip); statistics from a store-less launch of the same depthwise kernel.
        # Bit-identical to the un-fused schedule and 1.9x less HBM traffic, but MEASURED SLOWER on MI355X (level 0, 8 x 112^3:
        # 1.33 + 0.27 ms against 1.03 + 0.49 ms; profiles/r02_upfuse.txt): the mixer is bound by its VALU work (GELU), not by
        # HBM, and the prologue adds 20 % more of it -- off by default, kept as the switch for when the activation gets cheaper.
        self.fuse_up = os.environ.get("PYTC_FUSE_UP", "0") == "1"
        self.fuse_up_cin = (64, 128)       # input widths the fused up kernel is used for (A/B switch for measurements)
        # bf16 fused mixers at C_in <= 128: GroupNorm's affine folded into the expanding conv per sample (ops.groupnorm_fold_mlp in
        # the place of groupnorm_finalize), the mixer loads its operand raw.  Round 4: the mixers are bound by VALU issue and the
        # affine (unpack + fma + repack per element) was ~13 % of their instructions.  PYTC_FOLD_NORM=0 restores the affine prologue.
        self.fold_norm = os.environ.get("PYTC_FOLD_NORM", "1") != "0"
        # bf16 blocks of batches with at most this many voxel rows (levels 3-4 of an 8-window batch: 21 952 / 2 744 rows) run their
        # two 1x1x1 convs as two LDS-tiled GEMM launches (ops.pw_gemm) instead of the fused mixer: bit-identical, and the deep levels
        # stop leaving most SIMDs idle.  0 switches the path off.
        self.deep_gemm_rows = int(os.environ.get("PYTC_DEEP_GEMM_ROWS", "8192"))
        # The threshold is on the rows of ONE sample (round 5; it was N * rows <= 32768): the GEMM pair applies the GroupNorm affine to the
        # ACTIVATION, the fused mixer folds it into the expand WEIGHTS -- two rounding points -- so a rule that looked at the batch gave the
        # same window different bf16 bits alone (the engine's probe window, a ragged last batch) and inside a full batch,
        # breaking "chunked == whole volume" / "rank-sharded == single process".  8192 rows per sample reproduces the schedules the old rule
        # chose at the batch sizes it was measured at: MedNeXt-S 112^3 x 8 (level 3: 2 744 rows, level 4: 343) and MedNeXt-L 160^3 x 2
        # (level 3: 8 000, level 4: 1 000) take the GEMM pair, the 21 952 / 64 000 rows of their level 2 the fused mixer.
        # ... and so do blocks at least this wide whatever their row count (the 256 -> 512 -> 128 up block at 28^3); 0 = rows rule only
        self.deep_gemm_cin = int(os.environ.get("PYTC_DEEP_GEMM_CIN", "0"))
        # bf16 blocks of the mid-level shapes (ops.pw_mlp_lds_supported: 64->128->64, 128->256->64, 128->256->128, 64->128->32) with at
        # least this many voxel rows in the batch run the PERSISTENT mixer whose weight images stay in LDS (pw_mlp_lds_kernels.hip):
        # bit-identical, 15 ... 35 % faster at 8 windows; below the threshold staging the images costs more than it saves (5 x 14^3:
        # 27 against 19 us).  0 switches the path off.
        self.lds_mixer_rows = int(os.environ.get("PYTC_LDS_MIXER_ROWS", "131072"))
        # ... and blocks whose images exceed LDS (ops.pw_mlp_chunk_supported with c_hid >= PYTC_CHUNK_MIXER_HID: MedNeXt-L's 128->1024->128,
        # 256->2048->128, 128->512->64, 64->512->128) the kernel that streams the images through LDS one hidden chunk per workgroup step
        # (pw_mlp_chunk_kernels.hip; bit-identical) from this many rows in the batch.  0 switches the path off.
        self.chunk_mixer_rows = int(os.environ.get("PYTC_CHUNK_MIXER_ROWS", "16384"))
        self.chunk_mixer_hid = int(os.environ.get("PYTC_CHUNK_MIXER_HID", "512"))
        # bf16 residual blocks with 32 channels (level 0: the widest tensors of the network) can run as statistics pass + ONE fused kernel
        # (ops.dwmix): the depthwise output is re-formed in LDS and never stored -- 3 instead of 5 tensor passes per block, bit-identical
        # to the two-launch schedule.  Measured on MI355X (profiles/r05_fused_block.txt): the fused kernel is bound by instruction issue,
        # not by memory (44 % of its wave cycles are the mixer's GELU): 236 + 517 us against 318 + 14 + 412 us for 8 x 112^3 -- parity
        # (0.98x), 1.03x at MedNeXt-L's 2 x 160^3 x (32 -> 96 -> 32), whole step 6.70 against 6.54 ms per 8 windows.  Off by default.
        self.fuse_block = int(os.environ.get("PYTC_FUSE_BLOCK", "0"))

    # ---- parameter repacking (load time / after optimizer steps) -----------------------------
    def _taps(self, conv: nn.Module):
        w = conv.weight
        def make():
            c = w.shape[0]
            k = w.shape[-1]
            w3 = embed_stencil_2d(w.detach().float()) if w.dim() == 4 else w.detach().float()
            return w3.reshape(c, k ** 3).t().contiguous(), k
        return self.cache.get(("taps", id(conv)), [w], make)

    def _pw(self, conv: nn.Module, dt: torch.dtype, transposed: bool = False):
        w = conv.weight
        def make():
            w2 = w.detach().float().reshape(w.shape[0], w.shape[1]).contiguous()
            return ops.pw_pack_weight(w2, dt, transposed=transposed)
        return self.cache.get(("pw", id(conv), dt, transposed), [w], make)

    def _pw_paired(self, conv: nn.Module, transposed: bool = False, project: bool = False):
        """`project`: the mixer's second (projecting) conv -- its image is fp16 when ops.MLP_F16_PROJECT (packed-fp16 GELU +
        f16 MFMA in the fused mixer), bf16 otherwise."""
        w = conv.weight
        f16 = bool(project and ops.MLP_F16_PROJECT)
        def make():
            w2 = w.detach().float().reshape(w.shape[0], w.shape[1]).contiguous()
            return ops.pw_pack_weight_paired(w2, transposed=transposed, f16=f16)
        return self.cache.get(("pwp", id(conv), transposed, f16), [w], make)

    def _w2_matrix(self, conv: nn.Module):
        """(C_hid, C_in) fp32 contiguous view of the expanding 1x1x1 conv's weight (operand of ops.groupnorm_fold_mlp)."""
        w = conv.weight
        return self.cache.get(("w2mat", id(conv)), [w], lambda: w.detach().float().reshape(w.shape[0], w.shape[1]).contiguous())

    def _fold(self, m, st, count, C: int, c_hid: int):
        """-> (None, w2n, b2n) when the block's norm can be folded into its expanding conv, else (ab, shared image, b2)."""
        gamma, beta = self._vec(m.norm, "weight", m.norm.weight), self._vec(m.norm, "bias", m.norm.bias)
        if self.fold_norm and ops.groupnorm_fold_mlp_supported(C, c_hid):
            w2n, b2n = ops.groupnorm_fold_mlp(st, count, gamma, beta, m.norm.eps, self._w2_matrix(m.conv2),
                                              self._vec(m.conv2, "bias", m.conv2.bias))
            return None, w2n, b2n
        return (ops.groupnorm_finalize(st, count, gamma, beta, m.norm.eps), self._pw_paired(m.conv2),
                self._vec(m.conv2, "bias", m.conv2.bias))

    def _head_w(self, conv: nn.Module):
        """bf16 MFMA fragment image of the transposed 1x1x1 output conv (weights rounded like the un-fused head's)."""
        w = conv.weight
        def make():
            return ops.pack_head_fragment(w.detach().float().reshape(w.shape[0], w.shape[1]).t().contiguous())
        return self.cache.get(("headw", id(conv)), [w], make)

    def _grn(self, m, h: torch.Tensor, N: int, grid, c_hid: int, kind: str) -> torch.Tensor:
        """Global response normalisation of the expanded tensor (MedNeXtBlock.forward with grn=True):
        gx = ||h||_2 over space per (n, c), nx = gx / (mean_c gx + 1e-6), h <- gamma * (h * nx) + beta + h
        = h * (gamma * nx + 1) + beta: the column sums of squares come from the statistics kernel, the update is one
        per-(n, c) affine.  For the up block the padded front faces are not part of h's support."""
        Do, Ho, Wo = grid
        hv = h.view(N, Do, Ho, Wo, c_hid)
        core = hv[:, 1:, 1:, 1:].contiguous() if kind == "up" else hv
        sumsq = ops.channel_stats(core)[:, :, 1].sum(1)                       # (N, C)
        gx = sumsq.clamp_min(0).sqrt()
        nx = gx / (gx.mean(1, keepdim=True) + 1e-6)
        g = self._vec(m, "grn_gamma", m.grn_gamma).view(1, c_hid)
        b = self._vec(m, "grn_beta", m.grn_beta).view(1, c_hid)
        ab = torch.stack([g * nx + 1.0, b.expand(N, c_hid)], 1).contiguous()
        return ops.affine_act(hv, ab, nat.ACT_NONE, 0.0).view(N, Do * Ho * Wo, c_hid)

    def _vec(self, owner, name: str, p: Optional[torch.Tensor]):
        if p is None:
            return None
        return self.cache.get(("vec", id(owner), name), [p], lambda: _f32(p).reshape(-1))

    # ---- ops -----------------------------------------------------------------------------------
    def pointwise(self, x: torch.Tensor, conv: nn.Module, *, out_dtype=None, transposed=False, act=nat.ACT_NONE):
        """x (N, *spatial, C_in) -> (N, *spatial, C_out) through a 1x1x1 conv module."""
        N = x.shape[0]
        spatial = tuple(x.shape[1:-1])
        rows = 1
        for s in spatial:
            rows *= s
        c_in = x.shape[-1]
        c_out = conv.weight.shape[1] if transposed else conv.weight.shape[0]
        dt = x.dtype if x.dtype == torch.bfloat16 else torch.float32
        wdt = torch.bfloat16 if (dt == torch.bfloat16 or (out_dtype == torch.bfloat16)) else torch.float32
        wp = self._pw(conv, wdt, transposed)
        y = ops.pw_conv(x, wp, self._vec(conv, "bias", conv.bias), N=N, rows_per_sample=rows, c_in=c_in,
                        c_out=c_out, out_dtype=out_dtype or x.dtype, act=act)
        return y.view(N, *spatial, c_out)

    def block(self, m: MedNeXtBlock, x: torch.Tensor, skip: Optional[torch.Tensor] = None, head: Optional[nn.Module] = None,
              out: Optional[torch.Tensor] = None, proj: Optional[nn.Module] = None):
        """head: the network's output conv; when the fused mixer can carry it in its epilogue the block returns
        (None, logits fp32 (N, D, H, W, n_classes)) instead of its bf16 output.
        proj: a 32 -> 32 1x1x1 Conv3d applied to the block's output (the merged input projection of task heads): when the fused mixer
        can carry it the block returns (None, z bf16 (N, D, H, W, 32)) and its own output is never written; otherwise the block's output
        comes back as usual and the caller applies the conv.
        out: a dense buffer of the block's output shape (a sample slice of a batch tensor): the fused bf16 mixer writes its
        result there directly, every other schedule copies into it; the return value is `out` then (not with `head`).

        dim='2d' blocks run the same kernels on a depth-1 volume (N, 1, H, W, C): a k x k stencil is the centre z-plane of
        a k^3 one whose other planes only ever meet the zero padding, the stride-2 and 1x1 convs reduce to their 2-D forms
        at depth 1, and the statistics / GRN sums run over the same H x W support.  Only the up block needs care: the 3-D
        transposed conv + front padding doubles the depth as well, so its result is computed on a depth-2 grid whose
        plane 0 is the padded face and whose plane 1 is the 2-D answer."""
        if m.dim == "2d":
            if x.shape[1] != 1:
                raise ValueError(f"dim='2d' MedNeXt blocks take depth-1 volumes (N, 1, H, W, C), got {tuple(x.shape)}")
            if m.kind == "up":
                sk3 = None
                if skip is not None:
                    sk3 = skip.new_zeros((skip.shape[0], 2) + tuple(skip.shape[2:]))
                    sk3[:, 1] = skip[:, 0]
                y = self._block(m, x, sk3, None)[:, 1:2].contiguous()
                return y if out is None else out.copy_(y)
        return self._block(m, x, skip, head, out, proj)

    def _block(self, m: MedNeXtBlock, x: torch.Tensor, skip: Optional[torch.Tensor] = None, head: Optional[nn.Module] = None,
               out: Optional[torch.Tensor] = None, proj: Optional[nn.Module] = None):
        y = self._block_any(m, x, skip, head, out, proj)
        if out is None or isinstance(y, tuple) or y is out:
            return y
        if y.data_ptr() == out.data_ptr():          # written in place by the fused mixer (a view of `out`)
            return out
        return out.copy_(y.view(out.shape))

    def _block_any(self, m: MedNeXtBlock, x: torch.Tensor, skip: Optional[torch.Tensor] = None, head: Optional[nn.Module] = None,
                   out: Optional[torch.Tensor] = None, proj: Optional[nn.Module] = None):
        is_ln = isinstance(m.norm, _ChannelLayerNorm)
        if not is_ln and not isinstance(m.norm, nn.GroupNorm):
            raise NotImplementedError(f"unsupported MedNeXt norm module {type(m.norm).__name__}")
        dt = x.dtype
        N, D, H, W, C = x.shape
        taps, K = self._taps(m.conv1)
        b1 = self._vec(m.conv1, "bias", m.conv1.bias)
        kind = m.kind
        c_hid = m.conv2.weight.shape[0]
        c_out = m.conv3.weight.shape[0]
        if (kind == "up" and self.fused and self.fuse_up and C in self.fuse_up_cin and dt == torch.bfloat16 and K == 3 and not m.grn and not is_ln
                and m.dim == "3d"
                and skip is not None and m.conv2.bias is not None and m.conv3.bias is not None
                and ops.pw_mlp_up_supported(C, c_hid, c_out)):
            return self._up_block_fused(m, x, skip, taps, b1, c_hid, c_out)
        if (kind == "block" and self.fused and self.fuse_block and self.fold_norm and dt == torch.bfloat16 and K == 3 and not m.grn and not is_ln
                and m.dim == "3d" and m.conv2.bias is not None and m.conv3.bias is not None and ops.MLP_F16_PROJECT
                and ops.dwmix_supported(x, c_hid, c_out) and ops.groupnorm_fold_mlp_supported(C, c_hid)
                and proj is None
                and (head is None or (head.weight.shape[1] <= 16 and ops.pw_mlp_head_supported(C, c_hid, c_out)))):
            return self._block_dwmix(m, x, taps, b1, c_hid, c_out, head, out)
        if kind == "up":
            t, st = ops.dwconv3d(x, taps, b1, K=K, transposed=True, stats=not is_ln)
            count = float((2 * D - 1) * (2 * H - 1) * (2 * W - 1))
        else:
            t, st = ops.dwconv3d(x, taps, b1, K=K, stride=2 if kind == "down" else 1, stats=not is_ln)
            count = float(t.shape[1] * t.shape[2] * t.shape[3])
        gamma, beta = self._vec(m.norm, "weight", m.norm.weight), self._vec(m.norm, "bias", m.norm.bias)
        if is_ln:
            # channels-first LayerNorm: per-voxel statistics over C, applied by its own kernel (no (n, c) affine to fuse)
            t = ops.layernorm_rows(t, gamma, beta, m.norm.eps)
            ab = None
        else:
            ab = None                                   # finalized below: folded into conv2 on the fused bf16 path
        _, Do, Ho, Wo, _ = t.shape
        rows = Do * Ho * Wo
        # optional two-GEMM schedule for the deep levels (SMALL_ROWS_TWO_GEMMS, off by default: measured slower here): expand, then
        # project with the GELU in the operand prologue -- pw_fast shares out its output channels over blockIdx.z on small problems
        small = (self.fused and dt == torch.bfloat16 and not m.grn and not is_ln and N * rows < SMALL_ROWS_TWO_GEMMS
                 and ops.pw_conv_paired_supported(c_in=C, c_out=c_hid, in_dtype=dt, out_dtype=dt)
                 and ops.pw_conv_paired_supported(c_in=c_hid, c_out=c_out, in_dtype=dt, out_dtype=dt))
        if (not small and self.fused and dt == torch.bfloat16 and not m.grn and not is_ln and m.conv2.bias is not None
                and m.conv3.bias is not None and head is None and ops.MLP_F16_PROJECT
                and (rows <= self.deep_gemm_rows or (self.deep_gemm_cin and C >= self.deep_gemm_cin))
                and ops.pw_gemm_supported(C, c_hid) and ops.pw_gemm_supported(c_hid, c_out)):
            ab = ops.groupnorm_finalize(st, count, gamma, beta, m.norm.eps)
            return self._block_deep_gemm(m, x, t, ab, skip, (N, D, H, W, C), (Do, Ho, Wo), c_hid, c_out, out)
        if (not small and self.fused and dt == torch.bfloat16 and not m.grn and not is_ln and m.conv2.bias is not None
                and m.conv3.bias is not None and ops.pw_mlp_supported(C, c_hid, c_out)):
            ab, w2x, b2x = self._fold(m, st, count, C, c_hid)
            if (head is not None and kind == "block" and head.weight.shape[1] <= 16
                    and ops.pw_mlp_head_supported(C, c_hid, c_out)):
                _, logits = ops.pw_mlp_head(t, ab, w2x, b2x,
                                            self._pw_paired(m.conv3, project=True), self._vec(m.conv3, "bias", m.conv3.bias),
                                            self._head_w(head), self._vec(head, "bias", head.bias), N=N,
                                            rows_per_sample=rows, c_in=C, c_hid=c_hid, c_out=c_out,
                                            res=x if m.do_res else None, store_y=False)
                return None, logits.view(N, Do, Ho, Wo, -1)
            if (proj is not None and head is None and out is None and kind == "block" and ab is None and ops.MLP_F16_PROJECT
                    and tuple(proj.weight.shape[:2]) == (32, c_out) and ops.pw_mlp_proj_supported(C, c_hid, c_out, 32)):
                _, z = ops.pw_mlp_proj(t, w2x, b2x, self._pw_paired(m.conv3, project=True), self._vec(m.conv3, "bias", m.conv3.bias),
                                       self._pw_paired(proj), self._vec(proj, "bias", proj.bias), N=N, rows_per_sample=rows, c_in=C,
                                       c_hid=c_hid, c_out=c_out, res=x if m.do_res else None, store_y=False)
                return None, z.view(N, Do, Ho, Wo, 32)
            return self._block_fused(m, x, t, ab, skip, (N, D, H, W, C), (Do, Ho, Wo), c_hid, c_out, out, w2x, b2x)
        if not is_ln:
            ab = ops.groupnorm_finalize(st, count, gamma, beta, m.norm.eps)
        G = {}
        if small:
            h = ops.pw_conv(t, self._pw_paired(m.conv2), self._vec(m.conv2, "bias", m.conv2.bias), N=N, rows_per_sample=rows,
                            c_in=C, c_out=c_hid, out_dtype=dt, ab=ab, w_paired=True)                   # pre-activation
            w3, b3 = self._pw_paired(m.conv3), self._vec(m.conv3, "bias", m.conv3.bias)
            G = dict(pre_act=nat.ACT_GELU, w_paired=True)
        else:
            h = ops.pw_conv(t, self._pw(m.conv2, dt), self._vec(m.conv2, "bias", m.conv2.bias), N=N,
                            rows_per_sample=rows, c_in=C, c_out=c_hid, out_dtype=dt, ab=ab, act=nat.ACT_GELU)
            if m.grn:
                h = self._grn(m, h, N, (Do, Ho, Wo), c_hid, kind)
            w3, b3 = self._pw(m.conv3, dt), self._vec(m.conv3, "bias", m.conv3.bias)
        if kind == "block":
            y = ops.pw_conv(h, w3, b3, N=N, rows_per_sample=rows, c_in=c_hid, c_out=c_out, out_dtype=dt,
                            res=x if m.do_res else None, res_mode=nat.RES_ADD if m.do_res else nat.RES_NONE, **G)
        elif kind == "down":
            res = None
            if m.resample_do_res:
                res = ops.pw_conv(x, self._pw(m.res_conv, dt), self._vec(m.res_conv, "bias", m.res_conv.bias),
                                  N=N, rows_per_sample=rows, c_in=C, c_out=c_out, out_dtype=dt, gather=2,
                                  grid=(D, H, W))
            y = ops.pw_conv(h, w3, b3, N=N, rows_per_sample=rows, c_in=c_hid, c_out=c_out, out_dtype=dt,
                            res=res, res_mode=nat.RES_ADD if res is not None else nat.RES_NONE, **G)
        else:  # up: result + padded transposed-1x1 residual + encoder skip, one epilogue
            if skip is None:
                skip = torch.zeros((N, Do, Ho, Wo, c_out), dtype=dt, device=x.device)
            res_low = res_bias = None
            if m.resample_do_res:
                res_bias = self._vec(m.res_conv, "bias", m.res_conv.bias)
                res_low = ops.pw_conv(x, self._pw(m.res_conv, dt, transposed=True), res_bias, N=N,
                                      rows_per_sample=D * H * W, c_in=C, c_out=c_out, out_dtype=dt)
            y = ops.pw_conv(h, w3, b3, N=N, rows_per_sample=rows, c_in=c_hid, c_out=c_out, out_dtype=dt,
                            res=skip, res_mode=nat.RES_UPSAMPLE, grid=(Do, Ho, Wo), res_low=res_low,
                            res_bias=res_bias, **G)
        return y.view(N, Do, Ho, Wo, c_out)


def _block_dwmix(self, m, x, taps, b1, c_hid, c_out, head=None, out=None):
    """Residual block as statistics pass + fused kernel (HipBlockOps.fuse_block): same arithmetic as dwconv3d -> _fold -> pw_mlp /
    pw_mlp_head, the depthwise tensor never stored."""
    N, D, H, W, C = x.shape
    _, st = ops.dwconv3d(x, taps, b1, K=3, store=False)
    w2n, b2n = ops.groupnorm_fold_mlp(st, float(D * H * W), self._vec(m.norm, "weight", m.norm.weight), self._vec(m.norm, "bias", m.norm.bias),
                                      m.norm.eps, self._w2_matrix(m.conv2), self._vec(m.conv2, "bias", m.conv2.bias))
    w3, b3 = self._pw_paired(m.conv3, project=True), self._vec(m.conv3, "bias", m.conv3.bias)
    if head is not None:
        _, logits = ops.dwmix(x, taps, b1, w2n, b2n, w3, b3, c_hid=c_hid, c_out=c_out, residual=bool(m.do_res),
                              head_w=self._head_w(head), head_b=self._vec(head, "bias", head.bias), store_y=False)
        return None, logits
    if out is not None and tuple(out.shape) != (N, D, H, W, c_out):
        raise ValueError(f"block output buffer {tuple(out.shape)} != {(N, D, H, W, c_out)}")
    return ops.dwmix(x, taps, b1, w2n, b2n, w3, b3, c_hid=c_hid, c_out=c_out, residual=bool(m.do_res), y=out)


HipBlockOps._block_dwmix = None   # bound below


def _stem_block_fused(self, stem: nn.Module, m, x_cl: torch.Tensor, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Stem + first block without ever forming the stem output (bf16 inference, 1-channel fp32 input): the depthwise conv
    reads the network input (pytc_stem_dwconv3d_fwd), the mixer recomputes its residual from it (pytc_pw_mlp_stemres_fwd).
    Returns None when a precondition fails (the caller then runs stem and block separately)."""
    N, D, H, W, cin = x_cl.shape
    C = stem.weight.shape[0]
    K = m.conv1.weight.shape[-1]
    if (cin != 1 or x_cl.dtype != torch.float32 or stem.bias is None or m.kind != "block" or not m.do_res or m.grn
            or not isinstance(m.norm, nn.GroupNorm) or m.dim != "3d" or m.conv2.bias is None or m.conv3.bias is None
            or m.conv1.weight.shape[0] != C or W % 4 != 0 or not ops.stem_dwconv3d_supported(cin, C, K)):
        return None
    c_hid, c_out = m.conv2.weight.shape[0], m.conv3.weight.shape[0]
    if c_out != C or not ops.pw_mlp_supported(C, c_hid, c_out):
        return None
    # the un-fused stem kernel consumes bf16-rounded weights (bf16 storage path): same rounding here
    sw = self.cache.get(("stemw", id(stem)), [stem.weight], lambda: stem.weight.detach().float().reshape(-1).bfloat16().float().contiguous())
    sb = self._vec(stem, "bias", stem.bias)
    taps, _ = self._taps(m.conv1)
    packed = self.cache.get(("stemdw", id(stem), id(m.conv1)), [stem.weight, stem.bias, m.conv1.weight, m.conv1.bias],
                            lambda: ops.stem_dwconv3d_pack(sw, sb, taps, self._vec(m.conv1, "bias", m.conv1.bias)))
    t, st = ops.stem_dwconv3d(x_cl, packed)
    rows = D * H * W
    ab, w2x, b2x = self._fold(m, st, float(rows), C, c_hid)
    y = ops.pw_mlp_stemres(t, ab, w2x, b2x,
                           self._pw_paired(m.conv3, project=True), self._vec(m.conv3, "bias", m.conv3.bias), x_cl.reshape(N, rows), sw, sb,
                           N=N, rows_per_sample=rows, c_in=C, c_hid=c_hid, c_out=c_out,
                           y=None if out is None else out.view(N, rows, c_out))
    return y.view(N, D, H, W, c_out)


HipBlockOps._stem_block_fused = None   # bound below


def _block_fused(self, m, x, t, ab, skip, ishape, oshape, c_hid, c_out, out=None, w2=None, b2=None):
    """bf16 fast path: one pw_mlp launch per block (plus the tiny residual-conv GEMMs of down/up blocks).  (ab, w2, b2) from
    HipBlockOps._fold: ab None = the norm lives inside the per-sample expand operands."""
    N, D, H, W, C = ishape
    Do, Ho, Wo = oshape
    rows = Do * Ho * Wo
    dt = torch.bfloat16
    if w2 is None:
        w2, b2 = self._pw_paired(m.conv2), self._vec(m.conv2, "bias", m.conv2.bias)
    w3, b3 = self._pw_paired(m.conv3, project=True), self._vec(m.conv3, "bias", m.conv3.bias)
    kw = dict(N=N, rows_per_sample=rows, c_in=C, c_hid=c_hid, c_out=c_out)
    if (self.lds_mixer_rows and N * rows >= self.lds_mixer_rows and w3.dtype == torch.float16
            and ops.pw_mlp_lds_supported(C, c_hid, c_out)):
        kw["lds"] = True
    chunk_ok = (self.chunk_mixer_rows and N * rows >= self.chunk_mixer_rows and w3.dtype == torch.float16
                and ops.pw_mlp_chunk_supported(C, c_hid, c_out))
    if chunk_ok and (c_hid >= self.chunk_mixer_hid or (C, c_out) == (128, 64)):
        kw.pop("lds", None)                  # 128 -> 256 -> 64 (up block): 241 against 261 us for the LDS-resident form at 8 x 56^3
        kw["chunked"] = True
    if out is not None:
        if tuple(out.shape) != (N, Do, Ho, Wo, c_out):
            raise ValueError(f"block output buffer {tuple(out.shape)} != {(N, Do, Ho, Wo, c_out)}")
        kw["y"] = out.view(N, rows, c_out)
    if m.kind == "block":
        y = ops.pw_mlp(t, ab, w2, b2, w3, b3, res=x if m.do_res else None,
                       res_mode=nat.RES_ADD if m.do_res else nat.RES_NONE, **kw)
    elif m.kind == "down":
        res = None
        if m.resample_do_res:
            paired = ops.pw_conv_paired_supported(c_in=C, c_out=c_out, in_dtype=dt, out_dtype=dt, gather=2)
            wres = self._pw_paired(m.res_conv) if paired else self._pw(m.res_conv, dt)
            res = ops.pw_conv(x, wres, self._vec(m.res_conv, "bias", m.res_conv.bias), N=N, rows_per_sample=rows,
                              c_in=C, c_out=c_out, out_dtype=dt, gather=2, grid=(D, H, W), w_paired=paired)
        y = ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_ADD if res is not None else nat.RES_NONE, **kw)
    else:
        if skip is None:
            skip = torch.zeros((N, Do, Ho, Wo, c_out), dtype=dt, device=x.device)
        res_low = res_bias = None
        if m.resample_do_res:
            res_bias = self._vec(m.res_conv, "bias", m.res_conv.bias)
            paired = ops.pw_conv_paired_supported(c_in=C, c_out=c_out, in_dtype=dt, out_dtype=dt)
            wres = self._pw_paired(m.res_conv, transposed=True) if paired else self._pw(m.res_conv, dt, transposed=True)
            res_low = ops.pw_conv(x, wres, res_bias, N=N, rows_per_sample=D * H * W, c_in=C, c_out=c_out,
                                  out_dtype=dt, w_paired=paired)
        y = ops.pw_mlp(t, ab, w2, b2, w3, b3, res=skip, res_mode=nat.RES_UPSAMPLE, grid=(Do, Ho, Wo),
                       res_low=res_low, res_bias=res_bias, **kw)
    return y.view(N, Do, Ho, Wo, c_out)


def _block_deep_gemm(self, m, x, t, ab, skip, ishape, oshape, c_hid, c_out, out=None):
    """Deep-level schedule: expand (+ GroupNorm affine, GELU, fp16 hidden) and project (+ residual epilogue) as two GEMM launches."""
    N, D, H, W, C = ishape
    Do, Ho, Wo = oshape
    rows = Do * Ho * Wo
    dt = torch.bfloat16
    w2 = self.cache.get(("gw2", id(m.conv2)), [m.conv2.weight],
                        lambda: m.conv2.weight.detach().float().reshape(c_hid, C).to(torch.bfloat16).contiguous())
    w3 = self.cache.get(("gw3", id(m.conv3)), [m.conv3.weight],
                        lambda: m.conv3.weight.detach().float().reshape(c_out, c_hid).to(torch.float16).contiguous())
    b2, b3 = self._vec(m.conv2, "bias", m.conv2.bias), self._vec(m.conv3, "bias", m.conv3.bias)
    h = ops.pw_gemm(t.view(N, rows, C), w2, b2, N=N, rows_per_sample=rows, ab=ab, gelu=True)
    kw = dict(N=N, rows_per_sample=rows)
    if out is not None:
        kw["y"] = out.view(N, rows, c_out)
    if m.kind == "block":
        y = ops.pw_gemm(h, w3, b3, res=x if m.do_res else None, res_mode=nat.RES_ADD if m.do_res else nat.RES_NONE, **kw)
    elif m.kind == "down":
        res = None
        if m.resample_do_res:
            paired = ops.pw_conv_paired_supported(c_in=C, c_out=c_out, in_dtype=dt, out_dtype=dt, gather=2)
            wres = self._pw_paired(m.res_conv) if paired else self._pw(m.res_conv, dt)
            res = ops.pw_conv(x, wres, self._vec(m.res_conv, "bias", m.res_conv.bias), N=N, rows_per_sample=rows,
                              c_in=C, c_out=c_out, out_dtype=dt, gather=2, grid=(D, H, W), w_paired=paired)
        y = ops.pw_gemm(h, w3, b3, res=res, res_mode=nat.RES_ADD if res is not None else nat.RES_NONE, **kw)
    else:
        if skip is None:
            skip = torch.zeros((N, Do, Ho, Wo, c_out), dtype=dt, device=x.device)
        res_low = res_bias = None
        if m.resample_do_res:
            res_bias = self._vec(m.res_conv, "bias", m.res_conv.bias)
            paired = ops.pw_conv_paired_supported(c_in=C, c_out=c_out, in_dtype=dt, out_dtype=dt)
            wres = self._pw_paired(m.res_conv, transposed=True) if paired else self._pw(m.res_conv, dt, transposed=True)
            res_low = ops.pw_conv(x, wres, res_bias, N=N, rows_per_sample=D * H * W, c_in=C, c_out=c_out, out_dtype=dt, w_paired=paired)
        y = ops.pw_gemm(h, w3, b3, res=skip, res_mode=nat.RES_UPSAMPLE, grid=(Do, Ho, Wo), res_low=res_low, res_bias=res_bias, **kw)
    return y.view(N, Do, Ho, Wo, c_out)


def _up_block_fused(self, m, x, skip, taps, b1, c_hid, c_out):
    """Up block without the depthwise output in HBM: statistics-only pass of the transposed depthwise kernel (bit-identical
    partial sums), then ONE mixer launch that forms its operand from the low-resolution input."""
    N, D, H, W, C = x.shape
    dt = torch.bfloat16
    _, st = ops.dwconv3d(x, taps, b1, K=3, transposed=True, stats=True, store=False)
    ab = ops.groupnorm_finalize(st, float((2 * D - 1) * (2 * H - 1) * (2 * W - 1)), self._vec(m.norm, "weight", m.norm.weight),
                                self._vec(m.norm, "bias", m.norm.bias), m.norm.eps)
    res_low = res_bias = None
    if m.resample_do_res:
        res_bias = self._vec(m.res_conv, "bias", m.res_conv.bias)
        paired = ops.pw_conv_paired_supported(c_in=C, c_out=c_out, in_dtype=dt, out_dtype=dt)
        wres = self._pw_paired(m.res_conv, transposed=True) if paired else self._pw(m.res_conv, dt, transposed=True)
        res_low = ops.pw_conv(x, wres, res_bias, N=N, rows_per_sample=D * H * W, c_in=C, c_out=c_out, out_dtype=dt,
                              w_paired=paired)
    return ops.pw_mlp_up(x, taps, b1, ab, self._pw_paired(m.conv2), self._vec(m.conv2, "bias", m.conv2.bias),
                         self._pw_paired(m.conv3), self._vec(m.conv3, "bias", m.conv3.bias), skip.contiguous(),
                         c_hid=c_hid, c_out=c_out, res_low=res_low, res_bias=res_bias)


HipBlockOps._up_block_fused = _up_block_fused
HipBlockOps._block_deep_gemm = _block_deep_gemm
HipBlockOps._block_fused = _block_fused
HipBlockOps._stem_block_fused = _stem_block_fused
HipBlockOps._block_dwmix = _block_dwmix


def resolve_compute_dtype(module_pref: Optional[torch.dtype]) -> torch.dtype:
    """bf16 storage when requested on the module or under torch.autocast(bfloat16) (the reference's
    'bf16-mixed' Lightning precision, training/lightning/trainer.py:216-223); fp32 otherwise."""
    if module_pref is not None:
        return module_pref
    if torch.is_autocast_enabled("cuda"):
        adt = torch.get_autocast_dtype("cuda")
        if adt == torch.bfloat16:
            return torch.bfloat16
        if adt == torch.float16:
            raise RuntimeError("fp16 autocast is not supported by the MI355X engine; use bf16-mixed")
    return torch.float32


def embed_stencil_2d(w: torch.Tensor) -> torch.Tensor:
    """Depthwise (C, 1, k, k) stencil -> (C, 1, k, k, k) with the 2-D taps in the centre z-plane and zeros elsewhere."""
    k = w.shape[-1]
    out = w.new_zeros((w.shape[0], w.shape[1], k, k, k))
    out[:, :, k // 2] = w
    return out


def to_channels_last(x: torch.Tensor) -> torch.Tensor:
    """(N,C,D,H,W) -> contiguous (N,D,H,W,C); free when C == 1."""
    if x.shape[1] == 1:
        return x.reshape(x.shape[0], *x.shape[2:], 1)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_channels_first(y: torch.Tensor) -> torch.Tensor:
    """(N,D,H,W,C) -> (N,C,D,H,W); free when C == 1."""
    if y.shape[-1] == 1:
        return y.reshape(y.shape[0], 1, *y.shape[1:-1])
    return y.permute(0, 4, 1, 2, 3).contiguous()


class MedNeXt(nn.Module):
    """MedNeXt v1 trunk (Roy et al., MICCAI 2023) executed by hand-written gfx950 kernels."""

    def __init__(self, in_channels: int, n_channels: int, n_classes: int, exp_r: Union[int, Sequence[int]] = 4,
                 kernel_size: int = 7, enc_kernel_size: Optional[int] = None,
                 dec_kernel_size: Optional[int] = None, deep_supervision: bool = False, do_res: bool = False,
                 do_res_up_down: bool = False, checkpoint_style: Optional[str] = None,
                 block_counts: Sequence[int] = (2, 2, 2, 2, 2, 2, 2, 2, 2), norm_type: str = "group",
                 dim: str = "3d", grn: bool = False):
        super().__init__()
        if checkpoint_style not in (None, "outside_block"):
            raise ValueError("checkpoint_style must be None or 'outside_block'")
        if len(block_counts) != 9:
            raise ValueError("block_counts must have exactly 9 elements")
        self.do_ds = deep_supervision
        self.inside_block_checkpointing = False
        self.outside_block_checkpointing = checkpoint_style == "outside_block"
        # how the training path honours `outside_block` (training/autograd.py:use_block_recompute): 'auto' recomputes the
        # per-block activations only when keeping them would take more than half of the free HBM; 'always' / 'never' force it
        self.checkpoint_policy = "auto"
        if kernel_size is not None:
            enc_kernel_size = dec_kernel_size = kernel_size
        conv, _ = _conv_nd(dim)
        if isinstance(exp_r, int):
            exp_r = [exp_r] * 9
        exp_r = list(exp_r)
        n = n_channels
        self.dim = dim
        self.stem = conv(in_channels, n, kernel_size=1)

        def blocks(c, r, k, count):
            return nn.Sequential(*[MedNeXtBlock(c, c, r, k, do_res, norm_type, dim=dim, grn=grn)
                                   for _ in range(count)])

        for lvl in range(4):
            setattr(self, f"enc_block_{lvl}", blocks(n << lvl, exp_r[lvl], enc_kernel_size, block_counts[lvl]))
            setattr(self, f"down_{lvl}", MedNeXtDownBlock(n << lvl, n << (lvl + 1), exp_r[lvl + 1], enc_kernel_size,
                                                          do_res=do_res_up_down, norm_type=norm_type, dim=dim, grn=grn))
        self.bottleneck = blocks(n << 4, exp_r[4], dec_kernel_size, block_counts[4])
        for i, lvl in enumerate((3, 2, 1, 0)):
            setattr(self, f"up_{lvl}", MedNeXtUpBlock(n << (lvl + 1), n << lvl, exp_r[5 + i], dec_kernel_size,
                                                      do_res=do_res_up_down, norm_type=norm_type, dim=dim, grn=grn))
            setattr(self, f"dec_block_{lvl}", blocks(n << lvl, exp_r[5 + i], dec_kernel_size, block_counts[5 + i]))
        self.out_0 = OutBlock(n, n_classes, dim)
        self.dummy_tensor = nn.Parameter(torch.tensor([1.0]), requires_grad=True)
        if deep_supervision:
            for h in (1, 2, 3, 4):
                setattr(self, f"out_{h}", OutBlock(n << h, n_classes, dim))
        self.block_counts = list(block_counts)
        self.compute_dtype: Optional[torch.dtype] = None   # None -> follow autocast
        self._hip = HipBlockOps()
        self.fuse_head = True      # inference: output projection inside the last mixer's epilogue where a kernel exists
        # inference: stem folded into the first depthwise conv (406 us against 174 + 400 us for the two kernels it replaces,
        # the stem output never written) + the residual recomputed from the 1-channel input in that block's mixer
        self.fuse_stem = True
        self.l0_subbatch = L0_SUBBATCH   # bf16 inference: samples per depth-first chain at the full-resolution level (0 = off)

    # ---- engine ---------------------------------------------------------------------------------
    def _check_input(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("MedNeXt (pytorch_connectomics_amd) runs only on an MI355X/ROCm device: "
                               "there is no CPU path. Move the model and input to 'cuda'.")
        if self.dim == "2d":
            if x.dim() not in (4, 5) or (x.dim() == 5 and x.shape[2] != 1):
                raise ValueError(f"MedNeXt dim='2d' takes (B,C,H,W) (or (B,C,1,H,W)) inputs, got {tuple(x.shape)}")
        elif x.dim() != 5:
            raise ValueError(f"MedNeXt dim='3d' takes (B,C,D,H,W) inputs, got {tuple(x.shape)}")
        spatial = x.shape[-2:] if self.dim == "2d" else x.shape[2:]
        if any(int(v) % 16 for v in spatial):
            raise ValueError(f"MedNeXt needs spatial sizes divisible by 16, got {tuple(spatial)}")

    def _lift(self, x: torch.Tensor):
        """dim='2d': (B,C,H,W) -> the depth-1 volume (B,C,1,H,W) the kernels run on; -> (tensor, was_4d)."""
        if self.dim == "2d" and x.dim() == 4:
            return x.unsqueeze(2), True
        return x, False

    @staticmethod
    def _drop(y, was_4d: bool):
        if not was_4d:
            return y
        if isinstance(y, list):
            return [v.squeeze(2) for v in y]
        return y.squeeze(2)

    def features_cl(self, x_cl: torch.Tensor, collect: Optional[List[torch.Tensor]] = None,
                    head: Optional[nn.Module] = None, proj: Optional[nn.Module] = None):
        """Channels-last in (N,D,H,W,C_in) fp32/bf16 -> channels-last full-resolution features.  With `head` (the output
        conv) the last block may return (None, logits) instead, with `proj` (a 32 -> 32 conv of the features) (None, proj(features)) --
        see HipBlockOps.block."""
        hip = self._hip
        dt = resolve_compute_dtype(self.compute_dtype)
        spatial = x_cl.shape[2:4] if self.dim == "2d" else x_cl.shape[1:4]
        if any(int(v) % 16 for v in spatial) or (self.dim == "2d" and x_cl.shape[1] != 1):
            # four stride-2 stages: the decoder's skip additions need every level to halve exactly
            raise ValueError(f"MedNeXt needs spatial sizes divisible by 16, got {tuple(x_cl.shape[1:4])}"
                             + (" (dim='2d': depth must be 1)" if self.dim == "2d" else ""))
        enc0 = list(self.enc_block_0)
        N = int(x_cl.shape[0])
        sb = int(self.l0_subbatch)
        chained = 0 < sb < N and dt == torch.bfloat16 and hip.fused and self.dim == "3d"
        fuse_stem = self.fuse_stem and hip.fused and dt == torch.bfloat16 and bool(enc0)

        def level0_encoder(xs: torch.Tensor, skip_out: Optional[torch.Tensor], down_out: Optional[torch.Tensor]):
            """stem -> enc_block_0 -> down_0 on the samples of xs; the level's skip tensor and the down block's result go to the
            given buffers (sample slices of the batch tensors) when there are any."""
            blocks = list(enc0)
            h = None
            if fuse_stem:
                h = hip._stem_block_fused(self.stem, blocks[0], xs, skip_out if len(blocks) == 1 else None)
                if h is not None:
                    blocks = blocks[1:]          # stem output never written (None: not applicable)
            if h is None:
                h = hip.pointwise(xs, self.stem, out_dtype=dt)
                if not blocks and skip_out is not None:
                    h = skip_out.copy_(h)
            for bi, blk in enumerate(blocks):
                h = hip.block(blk, h, out=skip_out if bi == len(blocks) - 1 else None)
            return h, hip.block(self.down_0, h, out=down_out)

        def level0_decoder(xl: torch.Tensor, sk: torch.Tensor, y_out: Optional[torch.Tensor] = None):
            """up_0 -> dec_block_0 (the last block may carry the output head: -> (None, logits))."""
            h = hip.block(self.up_0, xl, skip=sk)
            blocks = list(self.dec_block_0)
            for bi, blk in enumerate(blocks):
                last = bi == len(blocks) - 1
                h = hip.block(blk, h, head=head if last else None, out=y_out if last else None, proj=proj if last else None)
            if not blocks and y_out is not None:
                h = y_out.copy_(h)
            return h

        skips = []
        if chained:
            _, D0, H0, W0, _ = x_cl.shape
            c0, c1 = self.stem.weight.shape[0], self.down_0.conv3.weight.shape[0]
            skip0 = torch.empty((N, D0, H0, W0, c0), dtype=dt, device=x_cl.device)
            x = torch.empty((N, D0 // 2, H0 // 2, W0 // 2, c1), dtype=dt, device=x_cl.device)
            for s0 in range(0, N, sb):
                level0_encoder(x_cl[s0:s0 + sb], skip0[s0:s0 + sb], x[s0:s0 + sb])
            skips.append(skip0)
        else:
            h, x = level0_encoder(x_cl, None, None)
            skips.append(h)
        for lvl in range(1, 4):
            for blk in getattr(self, f"enc_block_{lvl}"):
                x = hip.block(blk, x)
            skips.append(x)
            x = hip.block(getattr(self, f"down_{lvl}"), x)
        for blk in self.bottleneck:
            x = hip.block(blk, x)
        if collect is not None:
            collect.append(x)
        for lvl in (3, 2, 1):
            x = hip.block(getattr(self, f"up_{lvl}"), x, skip=skips[lvl])
            for blk in getattr(self, f"dec_block_{lvl}"):
                x = hip.block(blk, x)
            if collect is not None:
                collect.append(x)
        if not chained:
            return level0_decoder(x, skips[0])
        first = level0_decoder(x[0:sb], skips[0][0:sb])
        if isinstance(first, tuple):                    # (None, logits): 4 B per voxel and class, the concatenation is noise
            parts = [first[1]] + [level0_decoder(x[s0:s0 + sb], skips[0][s0:s0 + sb])[1] for s0 in range(sb, N, sb)]
            return None, torch.cat(parts, 0)
        y0 = torch.empty((N,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
        y0[0:sb].copy_(first)
        for s0 in range(sb, N, sb):
            level0_decoder(x[s0:s0 + sb], skips[0][s0:s0 + sb], y0[s0:s0 + sb])
        return y0

    def output_cl(self, feat_cl: torch.Tensor, head: int = 0) -> torch.Tensor:
        """features -> fp32 logits, channels-last."""
        return self._hip.pointwise(feat_cl, getattr(self, f"out_{head}").conv_out, out_dtype=torch.float32,
                                   transposed=True)

    def forward_cl(self, x_cl: torch.Tensor):
        """Channels-last entry used by the sliding-window engine: (N,D,H,W,C_in) -> (N,D,H,W,n_classes) fp32."""
        out = self.features_cl(x_cl, head=self.out_0.conv_out if self.fuse_head else None)
        if isinstance(out, tuple):          # the last mixer carried the output projection in its epilogue
            return out[1]
        return self.output_cl(out)

    # ---- reference-visible API (mednext_models.py:215-231) ---------------------------------------
    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input(x)
        x, flat = self._lift(x)
        return self._drop(to_channels_first(self.features_cl(to_channels_last(x.float())).float()), flat)

    def forward_output(self, features: torch.Tensor) -> torch.Tensor:
        if not features.is_cuda:
            raise RuntimeError("forward_output needs a CUDA tensor: there is no CPU path")
        dt = resolve_compute_dtype(self.compute_dtype)
        features, flat = self._lift(features)
        return self._drop(to_channels_first(self.output_cl(to_channels_last(features).to(dt))), flat)

    def _autograd_active(self) -> bool:
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def forward(self, x: torch.Tensor):
        self._check_input(x)
        x, flat = self._lift(x)
        return self._drop(self._forward5(x), flat)

    def _forward5(self, x: torch.Tensor):
        if self._autograd_active():
            # training step: forward AND backward run hand-written kernels (training/autograd.py)
            from ...training.autograd import mednext_train_forward
            out = mednext_train_forward(self, to_channels_last(x.float()), resolve_compute_dtype(self.compute_dtype))
            if isinstance(out, list):
                return [to_channels_first(o) for o in out]
            return to_channels_first(out)
        if not self.do_ds:
            return to_channels_first(self.forward_cl(to_channels_last(x.float())))
        feats: Optional[List[torch.Tensor]] = []
        f = self.features_cl(to_channels_last(x.float()), collect=feats)
        out = to_channels_first(self.output_cl(f))
        # feats = [bottleneck, dec_3, dec_2, dec_1] -> out_4 .. out_1 ; returned [x, ds_1 .. ds_4]
        ds = [to_channels_first(self.output_cl(ft, h)) for ft, h in zip(feats, (4, 3, 2, 1))]
        return [out, ds[3], ds[2], ds[1], ds[0]]


def create_mednext_v1(num_input_channels: int, num_classes: int, model_id: str, kernel_size: int = 3,
                      deep_supervision: bool = False) -> MedNeXt:
    """Size table of the upstream factory (S/B/M/L); see SURVEY.md section 8(c) for its provenance."""
    table = {
        "S": dict(exp_r=2, block_counts=[2] * 9, checkpoint_style=None),
        "B": dict(exp_r=[2, 3, 4, 4, 4, 4, 4, 3, 2], block_counts=[2] * 9, checkpoint_style=None),
        "M": dict(exp_r=[2, 3, 4, 4, 4, 4, 4, 3, 2], block_counts=[3, 4, 4, 4, 4, 4, 4, 4, 3],
                  checkpoint_style="outside_block"),
        "L": dict(exp_r=[3, 4, 8, 8, 8, 8, 8, 4, 3], block_counts=[3, 4, 8, 8, 8, 8, 8, 4, 3],
                  checkpoint_style="outside_block"),
    }
    if model_id not in table:
        raise ValueError(f"model_id must be one of S, B, M, L; got {model_id!r}")
    return MedNeXt(in_channels=num_input_channels, n_channels=32, n_classes=num_classes,
                   kernel_size=kernel_size, deep_supervision=deep_supervision, do_res=True,
                   do_res_up_down=True, **table[model_id])


__all__ = ["MedNeXt", "MedNeXtBlock", "MedNeXtDownBlock", "MedNeXtUpBlock", "OutBlock", "create_mednext_v1",
           "HipBlockOps", "to_channels_last", "to_channels_first", "resolve_compute_dtype"]
