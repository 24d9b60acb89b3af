"""GPU parity of the MedNeXt HIP forward against the CPU oracle (oracle/mednext_oracle.py; parity
unpinned w.r.t. the un-vendored nnunet_mednext package -- see DESIGN.md)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import mednext_oracle as MO

pytestmark = pytest.mark.gpu

# north_star tolerance: semantic/affinity maps within 1e-3 (fp32 path).  bf16 storage is a
# performance mode; its error budget (8-bit mantissa through ~40 residual blocks) is stated here.
TOL_F32_PROB = 1e-3
TOL_BF16_PROB = 4e-2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cfg(arch="mednext_custom", **mednext):
    return NS(model=NS(arch=NS(type=arch), in_channels=1, out_channels=2, mednext=NS(**mednext),
                       loss=NS(deep_supervision=mednext.pop("ds", False)), heads=None))


def _build(dev, *, n_channels, exp_r, kernel_size, block_counts, n_classes=2, ds=False, in_ch=1, seed=0):
    from pytorch_connectomics_amd.models.architectures.mednext import MedNeXt
    torch.manual_seed(seed)
    m = MedNeXt(in_ch, n_channels, n_classes, exp_r=exp_r, kernel_size=kernel_size, deep_supervision=ds,
                do_res=True, do_res_up_down=True, block_counts=block_counts)
    with torch.no_grad():   # non-trivial norm affine
        for p_name, p in m.named_parameters():
            if p_name.endswith("norm.weight"):
                p.add_(0.2 * torch.randn_like(p))
            if p_name.endswith("norm.bias"):
                p.add_(0.2 * torch.randn_like(p))
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to(dev).eval(), st


@pytest.mark.parametrize("n_channels,exp_r,k,counts,size", [
    (8, 2, 3, [1] * 9, 32),
    (16, [2, 3, 4, 4, 4, 4, 4, 3, 2], 3, [1, 2, 1, 1, 1, 1, 1, 2, 1], 32),
    (8, 2, 5, [1] * 9, 32),
    (4, 2, 3, [1] * 9, 16),      # channel counts below the vector width (scalar load path)
])
def test_mednext_fp32_matches_oracle(dev, n_channels, exp_r, k, counts, size):
    m, st = _build(dev, n_channels=n_channels, exp_r=exp_r, kernel_size=k, block_counts=counts)
    x = torch.randn(2, 1, size, size, size, generator=torch.Generator().manual_seed(1))
    ref = MO.forward(st, x, n_channels=n_channels, exp_r=exp_r, kernel_size=k, block_counts=counts)
    with torch.no_grad():
        got = m(x.to(dev)).cpu()
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert (torch.sigmoid(got) - torch.sigmoid(ref)).abs().max() < TOL_F32_PROB
    torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-3)
    # argmax labels bit-exact wherever the reference margin exceeds the tolerance
    margin = (ref[:, 0] - ref[:, 1]).abs() > 1e-3
    assert torch.equal(got.argmax(1)[margin], ref.argmax(1)[margin])


def test_mednext_s_fp32_and_bf16(dev):
    """The flagship topology (MedNeXt-S, k=3) on a 32^3 patch."""
    s = MO.SIZES["S"]
    m, st = _build(dev, n_channels=32, exp_r=2, kernel_size=3, block_counts=[2] * 9, n_classes=1)
    assert sum(p.numel() for p in m.parameters()) == 5_550_882
    x = torch.rand(2, 1, 32, 32, 32, generator=torch.Generator().manual_seed(2))
    ref = MO.forward(st, x, n_channels=32, exp_r=2, kernel_size=3, block_counts=[2] * 9)
    with torch.no_grad():
        got = m(x.to(dev)).cpu()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got16 = m(x.to(dev)).float().cpu()
    assert (torch.sigmoid(got) - torch.sigmoid(ref)).abs().max() < TOL_F32_PROB
    assert (torch.sigmoid(got16) - torch.sigmoid(ref)).abs().max() < TOL_BF16_PROB


def test_mednext_feature_contract_and_deep_supervision(dev):
    # reference tests/unit/test_mednext_features.py:26-55
    m, st = _build(dev, n_channels=8, exp_r=2, kernel_size=3, block_counts=[1] * 9, n_classes=3, ds=True)
    x = torch.randn(1, 1, 32, 32, 32).to(dev)
    with torch.no_grad():
        f = m.forward_features(x)
        outs = m(x)
        proj = m.forward_output(f)
    assert f.shape == (1, 8, 32, 32, 32)
    assert isinstance(outs, list) and len(outs) == 5
    assert [tuple(o.shape[2:]) for o in outs] == [(32,) * 3, (16,) * 3, (8,) * 3, (4,) * 3, (2,) * 3]
    assert torch.allclose(proj, outs[0], atol=1e-5)
    ref = MO.forward(st, x.cpu(), deep_supervision=True, n_channels=8, exp_r=2, kernel_size=3, block_counts=[1] * 9)
    for g, r in zip(outs, ref):
        torch.testing.assert_close(g.cpu(), r, rtol=1e-3, atol=1e-3)


def test_build_model_and_multihead(dev):
    from pytorch_connectomics_amd.models import build_model
    cfg = NS(model=NS(arch=NS(type="mednext_custom"), in_channels=1, out_channels=2,
                      mednext=NS(base_channels=8, exp_r=2, kernel_size=3, block_counts=[1] * 9),
                      loss=NS(deep_supervision=False),
                      heads={"aff": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
                             "sdt": {"out_channels": 1, "num_blocks": 0}}, primary_head="aff"))
    torch.manual_seed(0)
    model = build_model(cfg).to(dev).eval()
    x = torch.randn(1, 1, 32, 32, 32, device=dev)
    with torch.no_grad():
        out = model(x)
    assert set(out["output"]) == {"aff", "sdt"}
    assert out["output"]["aff"].shape == (1, 3, 32, 32, 32) and out["output"]["sdt"].shape == (1, 1, 32, 32, 32)
    # oracle: trunk features -> head blocks
    st = {k: v.detach().cpu() for k, v in model.model.state_dict().items()}
    f = MO.forward_features(st, x.cpu(), n_channels=8, exp_r=2, kernel_size=3, block_counts=[1] * 9)
    hs = {k: v.detach().cpu() for k, v in model.heads["aff"].state_dict().items()}
    import torch.nn.functional as F
    h = MO.block_forward(f, {("b." + k[len("blocks.0."):]): v for k, v in hs.items() if k.startswith("blocks.0.")}, "b", 3)
    ref = F.conv3d(h, hs["projection.weight"], hs["projection.bias"])
    torch.testing.assert_close(out["output"]["aff"].cpu(), ref, rtol=1e-3, atol=1e-3)


def test_sliding_window_with_mednext_matches_oracle(dev):
    """End-to-end hot path: on-device sliding window + MedNeXt forward vs oracle engine + oracle network."""
    from oracle import window_oracle as WO
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    from pytorch_connectomics_amd.models.architectures.mednext_models import MedNeXtWrapper
    m, st = _build(dev, n_channels=8, exp_r=2, kernel_size=3, block_counts=[1] * 9, n_classes=1)
    net = MedNeXtWrapper(m)
    vol = torch.rand(1, 1, 40, 48, 56, generator=torch.Generator().manual_seed(7))
    eng = EagerSlidingWindowEngine(roi_size=(32, 32, 32), sw_batch_size=4, overlap=0.5, mode="bump",
                                   padding_mode="constant", cval=0.0)
    got = eng(vol.to(dev), net).cpu()
    ref = WO.eager_sliding_window(vol, lambda x: MO.forward(st, x, n_channels=8, exp_r=2, kernel_size=3,
                                                            block_counts=[1] * 9),
                                  roi=(32, 32, 32), overlap=0.5, mode="bump", sw_batch_size=4)
    assert got.shape == ref.shape
    assert (torch.sigmoid(got) - torch.sigmoid(ref)).abs().max() < TOL_F32_PROB
    # also through the generic callable contract (NCDHW in / out)
    got2 = eng(vol.to(dev), lambda x: net(x)).cpu()
    assert torch.equal(got, got2)


@pytest.mark.parametrize("n_classes,size", [(1, "S"), (3, "S"), (2, "L")])
def test_fused_output_head_equals_unfused(n_classes, size):
    """The output projection carried in the last mixer's epilogue (pw_mlp_head) vs the separate head kernel on the
    stored bf16 block output: same bf16-rounded operands, fp32 sums in a different order."""
    from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1
    torch.manual_seed(n_classes)
    m = create_mednext_v1(1, n_classes, size, 3).cuda().eval()
    m.compute_dtype = torch.bfloat16
    with torch.no_grad():
        m.out_0.conv_out.bias.normal_()
        x = torch.randn(2, 16, 32, 48, 1, device="cuda")
        m.fuse_head = False
        ref = m.forward_cl(x)
        m.fuse_head = True
        got = m.forward_cl(x)
        assert got.shape == ref.shape == (2, 16, 32, 48, n_classes) and got.dtype == torch.float32
        with pytest.raises(ValueError, match="divisible by 16"):
            m.forward_cl(torch.randn(1, 24, 32, 32, 1, device="cuda"))
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-6 * scale + 1e-6
        y1 = m(x.permute(0, 4, 1, 2, 3))
        torch.testing.assert_close(y1, got.permute(0, 4, 1, 2, 3))
        assert torch.equal(m.forward_cl(x), got)           # deterministic


@pytest.mark.parametrize("shape", [(16, 32, 48), (32, 16, 16)])
def test_fused_stem_equals_unfused_and_oracle_math(shape):
    """Stem folded into the first depthwise conv (stem output never written) + residual recomputed in the mixer, against
    the un-fused kernels (which round the stem output to bf16 first) and the kernel itself against fp64 math."""
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1
    torch.manual_seed(7)
    m = create_mednext_v1(1, 2, "S", 3).cuda().eval()
    m.compute_dtype = torch.bfloat16
    with torch.no_grad():
        m.stem.bias.normal_(0, 0.5)
        x = torch.randn(2, *shape, 1, device="cuda")
        m.fuse_stem = False
        ref = m.forward_cl(x)
        m.fuse_stem = True
        got = m.forward_cl(x)
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 4e-2 * scale          # bf16 storage path tolerance
        assert float((got - ref).abs().mean()) <= 4e-3 * scale
        assert torch.equal(m.forward_cl(x), got)
        # the fused depthwise kernel alone vs float64: t = dwconv3(stem(x)), zero padding of the stem output
        blk = m.enc_block_0[0]
        sw = m.stem.weight.detach().float().reshape(-1)
        sb = m.stem.bias.detach().float()
        taps = blk.conv1.weight.detach().float().reshape(32, 27).t().contiguous()
        t, st = ops.stem_dwconv3d(x, ops.stem_dwconv3d_pack(sw, sb, taps, blk.conv1.bias.detach().float()))
        x64 = x.double().cpu().permute(0, 4, 1, 2, 3)
        s64 = torch.nn.functional.conv3d(x64, m.stem.weight.detach().double().cpu(), m.stem.bias.detach().double().cpu())
        t64 = torch.nn.functional.conv3d(s64, blk.conv1.weight.detach().double().cpu(), blk.conv1.bias.detach().double().cpu(),
                                         padding=1, groups=32).permute(0, 2, 3, 4, 1)
        err = (t.double().cpu() - t64).abs().max() / t64.abs().max()
        assert float(err) < 6e-3                                         # one bf16 rounding of the stored result
        tb = t.float()
        torch.testing.assert_close(st[:, :, 0].sum(1).cpu(), tb.sum((1, 2, 3)).cpu(), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(st[:, :, 1].sum(1).cpu(), (tb * tb).sum((1, 2, 3)).cpu(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("norm_type,grn", [("group", True), ("layer", False), ("layer", True)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_mednext_grn_and_layernorm_variants_match_oracle(dev, norm_type, grn, dt):
    """build_mednext_custom's norm_type='layer' (channels-first LayerNorm) and grn=True (global response norm) --
    mednext_models.py:449-463 constructor arguments -- inference against the oracle."""
    from pytorch_connectomics_amd.models.architectures.mednext import MedNeXt
    torch.manual_seed(5)
    kw = dict(n_channels=8, exp_r=[2, 3, 2, 2, 2, 2, 2, 3, 2], kernel_size=3, block_counts=[1, 1, 1, 1, 1, 1, 1, 1, 2])
    m = MedNeXt(1, kw["n_channels"], 2, exp_r=kw["exp_r"], kernel_size=3, do_res=True, do_res_up_down=True,
                block_counts=kw["block_counts"], norm_type=norm_type, grn=grn)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("norm.weight") or n.endswith("norm.bias"):
                p.add_(0.2 * torch.randn_like(p))
            if "grn_" in n:
                p.normal_(0, 0.3)
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 1, 32, 16, 32, generator=torch.Generator().manual_seed(6))
    ref = MO.forward(st, x, norm_type=norm_type, grn=grn, **kw)
    m = m.to(dev).eval()
    m.compute_dtype = dt
    with torch.no_grad():
        got = m(x.to(dev)).cpu()
    assert got.shape == ref.shape
    tol = TOL_F32_PROB if dt == torch.float32 else TOL_BF16_PROB
    assert (torch.sigmoid(got) - torch.sigmoid(ref)).abs().max() < tol
    if dt == torch.float32:
        torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)
    tr = m.train()(x.to(dev))                      # the variants train too (NormVariantBlockFn; gradients: test_gpu_training.py)
    assert tr.requires_grad
    tol_train = 1e-3 if dt == torch.float32 else TOL_BF16_PROB
    assert (torch.sigmoid(tr.detach().float().cpu()) - torch.sigmoid(ref)).abs().max() < tol_train


@pytest.mark.parametrize("size,shape", [("S", (32, 32, 48)), ("S", (16, 48, 80)), ("L", (32, 32, 32)), ("B", (16, 32, 64))])
def test_fused_up_block_is_bit_identical_to_unfused(size, shape, monkeypatch):
    """pw_mlp_up (depthwise transposed conv recomputed in the mixer prologue, statistics from the store-less launch of the same
    depthwise kernel) against dwconvT3d + pw_mlp(RES_UPSAMPLE): same fp32 operation order, same bf16 roundings, same partial-sum
    tree -> bit-identical logits.  Shapes include low-resolution rows that are not multiples of the 16-cell tile.  The fused up
    kernel (off by default: measured slower) keeps the bf16 hidden activation, so the comparison runs the mixers in that mode."""
    from pytorch_connectomics_amd import hip_ops
    from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1
    monkeypatch.setattr(hip_ops, "MLP_F16_PROJECT", False)
    torch.manual_seed(3)
    m = create_mednext_v1(1, 2, size, 3).cuda().eval()
    m.compute_dtype = torch.bfloat16
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n.endswith("norm.weight") or n.endswith("norm.bias") or n.endswith("conv1.bias") or n.endswith("res_conv.bias"):
                p_.add_(0.2 * torch.randn_like(p_))
        x = torch.randn(2, *shape, 1, device="cuda")
        m._hip.fold_norm = False          # the fused up kernel applies the affine itself: compare like with like
        m._hip.fuse_up = False
        ref = m.forward_cl(x)
        m._hip.fuse_up = True
        got = m.forward_cl(x)
        assert torch.equal(got, ref)
        m._hip.fuse_up_cin = (64,)            # level 0 only
        assert torch.equal(m.forward_cl(x), ref)


def test_norm_folded_mixers_stay_within_the_affine_form_of_the_oracle_error(dev):
    """HipBlockOps.fold_norm (round 4, default): GroupNorm's affine inside per-sample expand weights instead of the mixer prologue.
    Same arithmetic, one rounding moved (weight instead of normalised activation): against the fp32 path of the same weights the
    folded bf16 forward is no further away than the affine-prologue form (10 % slack), at every level a MedNeXt-S forward has."""
    m, _ = _build(dev, n_channels=32, exp_r=2, kernel_size=3, block_counts=[2] * 9, ds=False)
    x = torch.rand(2, 32, 48, 48, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    with torch.no_grad():
        m.compute_dtype = torch.float32
        ref = m.forward_cl(x).float()
        m.compute_dtype = torch.bfloat16
        errs = {}
        for fold in (False, True):
            m._hip.fold_norm = fold
            errs[fold] = float((m.forward_cl(x).float() - ref).abs().mean())
    assert errs[True] <= 1.10 * errs[False] + 1e-6, errs


@pytest.mark.parametrize("counts,ds", [([2] * 9, False), ([1, 2, 1, 1, 1, 1, 1, 2, 1], True), ([1] * 9, False)])
def test_level0_subbatch_is_bit_identical(dev, counts, ds):
    """MedNeXt.l0_subbatch (PYTC_L0_SUBBATCH): the full-resolution level run depth-first over sample slices -- stem + encoder
    blocks + down block, later up block + decoder blocks (+ head) per slice, results written straight into the batch tensors --
    must not change a single bit of the output, for slices that divide the batch and for a ragged last slice, with the output
    head in the last mixer's epilogue, without it, and with deep supervision."""
    m, _ = _build(dev, n_channels=32, exp_r=2, kernel_size=3, block_counts=counts, ds=ds)
    m.compute_dtype = torch.bfloat16
    x = torch.randn(5, 1, 32, 32, 48, generator=torch.Generator().manual_seed(3)).to(dev)

    def run():
        with torch.no_grad():
            out = m(x)
            feats = m.forward_features(x)
        outs = out if isinstance(out, list) else [out]
        return [o.clone() for o in outs] + [feats.clone()]

    m.l0_subbatch = 0
    ref = run()
    for sb in (1, 2, 4, 5, 8):
        m.l0_subbatch = sb
        got = run()
        assert len(got) == len(ref)
        for a, b in zip(ref, got):
            assert a.shape == b.shape and a.dtype == b.dtype
            assert torch.equal(a, b), f"l0_subbatch={sb}: max |d| {(a.float() - b.float()).abs().max().item():.3e}"
    m.fuse_head = False                # output projection as its own launch: the chains end in the block's bf16 output
    m.l0_subbatch = 0
    ref = run()
    m.l0_subbatch = 2
    for a, b in zip(ref, run()):
        assert torch.equal(a, b)


def test_a_window_gives_the_same_bits_alone_and_inside_a_batch_at_112():
    """the block schedule (fused folded mixer / LDS-resident mixer / two-GEMM pair of the deep levels) is chosen from
    batch-independent quantities, so a 112^3 window -- whose levels straddle every row threshold at N = 1 .. 8 -- gives the same bf16
    bits as the engine's probe window (N = 1), in a ragged last batch (N = 3) and in a full batch of 8."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.models import build_model
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=1, mednext=NS(size="S", kernel_size=3),
                      loss=NS(deep_supervision=False), heads=None))
    torch.manual_seed(0)
    model = build_model(cfg).cuda().eval()
    model.model.compute_dtype = torch.bfloat16
    x = torch.rand(8, 112, 112, 112, 1, device="cuda")
    with torch.no_grad():
        full = model.forward_cl(x).clone()
        one = model.forward_cl(x[5:6].contiguous()).clone()
        three = model.forward_cl(x[2:5].contiguous()).clone()
    assert torch.equal(one[0], full[5])
    assert torch.equal(three, full[2:5])


def test_narrow_task_heads_merged_into_one_block_diagonal_head_match_the_per_head_passes(monkeypatch):
    """MedNeXtMultiHeadWrapper._merged_heads: three 8-channel MitoEM heads (tutorials/mitoEM/common.yaml:10-39) as one 32-channel head
    with block-diagonal weights.  Same function as the per-head passes -- per-channel GroupNorm, zero off-diagonal blocks -- on other
    kernels (matrix-core depthwise conv, MFMA mixer): equal to the rounding of the bf16 path, far inside the C4 oracle gate."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.models import build_model
    from pytorch_connectomics_amd.models.architectures import mednext_models as MM
    heads = {"aff_r1": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
             "aff_r5": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
             "sdt": {"out_channels": 1, "num_blocks": 1, "hidden_channels": 8}}
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=7, mednext=NS(size="L", kernel_size=3),
                      loss=NS(deep_supervision=False), heads=heads, primary_head="aff_r1"))
    torch.manual_seed(0)
    model = build_model(cfg).cuda().eval()
    with torch.no_grad():                                  # heads are zero-initialised biases by default: give every parameter a value
        for p in model.heads.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    model.model.compute_dtype = torch.bfloat16
    x = torch.rand(2, 32, 32, 48, 1, device="cuda")
    with torch.no_grad():
        feat = model.model.features_cl(x)
        monkeypatch.setattr(MM, "MERGE_NARROW_HEADS", False)
        want = model.forward_heads_cl(feat)
        monkeypatch.setattr(MM, "MERGE_NARROW_HEADS", True)
        assert model._merged_heads(feat) is not None
        got = model.forward_heads_cl(feat)
        cat = model._merged_heads_cl(feat)
    assert list(got) == list(want) == ["aff_r1", "aff_r5", "sdt"] and cat.shape[-1] == 7
    for k in want:
        scale = float(want[k].abs().max())
        err = (got[k] - want[k]).abs()
        assert got[k].shape == want[k].shape and float(err.max()) < 0.03 * scale and float(err.mean()) < 3e-3 * scale, (k, float(err.max()), scale)
    # the merged out-projection in the last mixer's epilogue (default) against its own launch: same bf16 block output, same weights
    with torch.no_grad():
        monkeypatch.setattr(MM, "FUSE_MERGED_HEAD_PROJECTION", False)
        unfused = model._merged_heads_cl(feat)
        monkeypatch.setattr(MM, "FUSE_MERGED_HEAD_PROJECTION", True)
    assert unfused.dtype == cat.dtype == torch.float32 and unfused.shape == cat.shape
    torch.testing.assert_close(cat, unfused, rtol=1e-5, atol=1e-5 * float(unfused.abs().max()))
    # the whole channels-last forward: the merged INPUT projection in the epilogue of the trunk's last mixer (pytc_pw_mlp_proj_fwd: the
    # features are never written) against features -> projection launch; one bf16 rounding of z may fall differently (bias-first accumulate)
    with torch.no_grad():
        whole = model.forward_cl(x)
        monkeypatch.setattr(MM, "FUSE_MERGED_HEAD_PROJECTION", False)
        whole_unfused = model.forward_cl(x)
        monkeypatch.setattr(MM, "FUSE_MERGED_HEAD_PROJECTION", True)
    assert whole.shape == whole_unfused.shape == cat.shape and whole.dtype == torch.float32
    scale = float(whole_unfused.abs().max())
    err = (whole - whole_unfused).abs()
    assert float(err.max()) < 0.01 * scale and float(err.mean()) < 5e-4 * scale, (float(err.max()), float(err.mean()), scale)
    from pytorch_connectomics_amd import hip_ops as ops
    with ops.profiled() as prof, torch.no_grad():                # ... and it is that kernel which ran
        model.forward_cl(x)
    torch.cuda.synchronize()
    assert any(str(r[5]) == "pw_mlp_dma_kernel<3, 3>" for r in prof.records), sorted({str(r[5]) for r in prof.records})
    # a head parameter changes -> the merged weights are rebuilt
    with torch.no_grad():
        model.heads["sdt"].projection.bias.add_(1.0)
        again = model.forward_heads_cl(feat)
    assert float((again["sdt"] - got["sdt"] - 1.0).abs().max()) < 1e-5 and torch.equal(again["aff_r1"], got["aff_r1"])
    assert "_merged_cache" not in dict(model.named_modules()) and not any("merged" in k for k in model.state_dict())
