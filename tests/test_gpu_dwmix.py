"""Fused MedNeXt residual block (pytc_dwmix_fwd, csrc/dwconv_mfma_kernels.hip MIXHC > 0): statistics-only pass + one kernel that
re-forms the depthwise conv in LDS and runs the channel mixer on it.  The contract is BIT-IDENTITY with the two-launch schedule
(pytc_dwconv3d_fwd -> pytc_groupnorm_fold_mlp -> pytc_pw_mlp_fwd / pytc_pw_mlp_head_fwd) on the same operands, on ragged footprints,
ragged z-chunks, every hidden width, with and without the residual, with the output head in the epilogue; and at the network level
the same bits with the path on and off."""
import pytest
import torch

pytestmark = pytest.mark.gpu

dev = torch.device("cuda")
bf = torch.bfloat16


def _operands(N, D, H, W, c_hid, seed, scale=1.0, offset=0.0):
    from pytorch_connectomics_amd import hip_ops as ops
    g = torch.Generator(device=dev).manual_seed(seed)
    x = (torch.randn(N, D, H, W, 32, device=dev, generator=g) * scale + offset).to(bf)
    taps = torch.randn(27, 32, device=dev, generator=g) * 0.2
    b1 = torch.randn(32, device=dev, generator=g) * 0.3
    gamma = torch.rand(32, device=dev, generator=g) + 0.5
    beta = torch.randn(32, device=dev, generator=g) * 0.2
    w2 = (torch.randn(c_hid, 32, device=dev, generator=g) / 32 ** 0.5).contiguous()
    b2 = torch.randn(c_hid, device=dev, generator=g) * 0.1
    w3_clear = (torch.randn(32, c_hid, device=dev, generator=g) / c_hid ** 0.5).contiguous()
    w3 = ops.pw_pack_weight_paired(w3_clear, f16=True)
    b3 = torch.randn(32, device=dev, generator=g) * 0.1
    _operands.w3_clear = w3_clear            # the un-packed projection weights of the last call (the fp32 restatement reads them)
    return x, taps, b1, gamma, beta, w2, b2, w3, b3


def _two_launch(ops, nat, x, taps, b1, gamma, beta, w2, b2, w3, b3, c_hid, residual, head=None):
    N, D, H, W, _ = x.shape
    t, st = ops.dwconv3d(x, taps, b1, K=3)
    w2n, b2n = ops.groupnorm_fold_mlp(st, float(D * H * W), gamma, beta, 1e-5, w2, b2)
    kw = dict(N=N, rows_per_sample=D * H * W, c_in=32, c_hid=c_hid, c_out=32)
    if head is not None:
        _, logits = ops.pw_mlp_head(t, None, w2n, b2n, w3, b3, head[0], head[1], res=x if residual else None, store_y=False, **kw)
        return logits.view(N, D, H, W, -1)
    y = ops.pw_mlp(t, None, w2n, b2n, w3, b3, res=x if residual else None, res_mode=nat.RES_ADD if residual else nat.RES_NONE, **kw)
    return y.view(N, D, H, W, 32)


def _fused(ops, x, taps, b1, gamma, beta, w2, b2, w3, b3, c_hid, residual, head=None):
    N, D, H, W, _ = x.shape
    y0, st = ops.dwconv3d(x, taps, b1, K=3, store=False)
    assert y0 is None
    w2n, b2n = ops.groupnorm_fold_mlp(st, float(D * H * W), gamma, beta, 1e-5, w2, b2)
    if head is not None:
        y, logits = ops.dwmix(x, taps, b1, w2n, b2n, w3, b3, c_hid=c_hid, residual=residual, head_w=head[0], head_b=head[1], store_y=False)
        assert y is None
        return logits
    return ops.dwmix(x, taps, b1, w2n, b2n, w3, b3, c_hid=c_hid, residual=residual)


@pytest.mark.parametrize("shape,c_hid,residual,variant", [
    ((2, 24, 40, 33), 64, True, 0),        # ragged x footprints, two z-chunks
    ((1, 9, 17, 16), 64, True, 0),         # one short chunk, ragged y
    ((3, 30, 16, 25), 96, True, 0),        # MedNeXt-L's level-0 width (exp_r 3)
    ((1, 16, 24, 24), 128, False, 0),      # exp_r 4, no residual
    ((2, 19, 32, 16), 64, True, 1),        # hi + lo depthwise weights
    ((2, 8, 12, 10), 64, True, 0),         # small planes (8 <= H, W < 16): always hi + lo
    ((1, 43, 56, 48), 64, False, 0),
])
def test_fused_block_is_bit_identical_to_the_two_launch_schedule(shape, c_hid, residual, variant):
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    N, D, H, W = shape
    assert ops.dwmix_supported(torch.empty(N, D, H, W, 32, device=dev, dtype=bf), c_hid, 32)
    opnds = _operands(N, D, H, W, c_hid, seed=D * 7 + W)
    ops.set_tuning("dwconv_mfma_variant", variant)
    try:
        want = _two_launch(ops, nat, *opnds, c_hid, residual)
        got = _fused(ops, *opnds, c_hid, residual)
    finally:
        ops.set_tuning("dwconv_mfma_variant", 0)
    torch.cuda.synchronize()
    assert torch.isfinite(want.float()).all()
    assert torch.equal(got, want), f"max |diff| {float((got.float() - want.float()).abs().max())}"


def test_fused_block_with_the_output_head_and_batch_invariance():
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    N, D, H, W, c_hid = 3, 20, 24, 40, 64
    opnds = _operands(N, D, H, W, c_hid, seed=5)
    g = torch.Generator(device=dev).manual_seed(11)
    head = (ops.pack_head_fragment(torch.randn(3, 32, device=dev, generator=g) * 0.3), torch.randn(3, device=dev, generator=g))
    want = _two_launch(ops, nat, *opnds, c_hid, True, head)
    got = _fused(ops, *opnds, c_hid, True, head)
    assert got.shape == (N, D, H, W, 3) and torch.equal(got, want)
    # a sample gives the same bits alone and inside a batch (per-sample statistics, batch-invariant z-chunks)
    x = opnds[0]
    alone = _fused(ops, x[1:2].contiguous(), *opnds[1:], c_hid, True)
    batch = _fused(ops, *opnds, c_hid, True)
    assert torch.equal(alone[0], batch[1])


def test_fused_block_against_an_fp32_restatement():
    """The two-launch schedule is itself tested against fp64 / fp32 references kernel by kernel; this pins the fused path on its own:
    depthwise conv (bf16 taps, fp32 accumulation, bf16 rounding) -> instance norm over the volume -> expand -> exact GELU -> project
    -> + x in fp32 torch, within the bf16 path's rounding class."""
    import torch.nn.functional as F
    from pytorch_connectomics_amd import hip_ops as ops
    N, D, H, W, c_hid = 2, 16, 24, 24, 64
    x, taps, b1, gamma, beta, w2, b2, w3p, b3 = _operands(N, D, H, W, c_hid, seed=3)
    w3 = _operands.w3_clear
    got = _fused(ops, x, taps, b1, gamma, beta, w2, b2, w3p, b3, c_hid, True).float()
    xf = x.float().permute(0, 4, 1, 2, 3)
    t = F.conv3d(xf, taps.to(bf).float().t().reshape(32, 1, 3, 3, 3), b1, padding=1, groups=32).to(bf).float()
    tn = F.group_norm(t, 32, gamma, beta, 1e-5)
    h = F.gelu(F.conv3d(tn, w2.view(c_hid, 32, 1, 1, 1), b2))
    want = (F.conv3d(h, w3.view(32, c_hid, 1, 1, 1), b3) + xf).permute(0, 2, 3, 4, 1)
    err = (got - want).abs()
    assert float(err.max()) < 0.08 * float(want.abs().max()) and float(err.mean()) < 6e-3 * float(want.abs().mean())


def test_mednext_forward_is_bit_identical_with_the_fused_block_on_and_off(monkeypatch):
    """MedNeXt-S (exp_r 2) and a exp_r-3 trunk: forward_cl with the fused block on (HipBlockOps.fuse_block = 1; default 0) equals the default schedule bit for bit (level-0 blocks
    and the head-carrying last block run fused; everything else is untouched)."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd.models import build_model
    for size in ("S", "L"):
        cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=2, mednext=NS(size=size, kernel_size=3),
                          loss=NS(deep_supervision=False), heads=None))
        torch.manual_seed(0)
        model = build_model(cfg).to(dev).eval()
        model.model.compute_dtype = bf
        x = torch.rand(2, 32, 48, 32, 1, device=dev)
        outs = {}
        for flag in (0, 1):
            # the switch is read when the block-ops object is built: set it on the instance the model holds
            for obj in _block_ops_objects(model):
                obj.fuse_block = flag
            with ops.profiled() as prof, torch.no_grad():
                outs[flag] = model.forward_cl(x).clone()
            labels = set(prof.summary())
            assert any(k.startswith("dwmix_fwd") for k in labels) == bool(flag), labels
        assert torch.equal(outs[0], outs[1])


def _block_ops_objects(model):
    from pytorch_connectomics_amd.models.architectures.mednext import HipBlockOps
    seen = []
    for mod in model.modules():
        for v in vars(mod).values():
            if isinstance(v, HipBlockOps) and v not in seen:
                seen.append(v)
    assert seen, "no HipBlockOps instance found on the model"
    return seen
