"""MedNeXt dim='2d' (reference constructor mednext_models.py:449-467, used by tutorials/mito_mitolab.yaml): the 2-D network
runs the 3-D kernels on depth-1 volumes (models/architectures/mednext.py HipBlockOps.block).  Forward, backward, heads, TTA
against the CPU oracle's Conv2d twin (oracle/mednext_oracle.py, dim='2d')."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn.functional as F

from oracle import mednext_oracle as MO

pytestmark = pytest.mark.gpu


def _build(*, n_channels, k=3, counts=(1,) * 9, n_classes=2, ds=False, norm_type="group", grn=False, exp_r=2, seed=0):
    from pytorch_connectomics_amd.models.architectures.mednext import MedNeXt
    torch.manual_seed(seed)
    m = MedNeXt(1, n_channels, n_classes, exp_r=exp_r, kernel_size=k, deep_supervision=ds, do_res=True, do_res_up_down=True,
                block_counts=list(counts), norm_type=norm_type, dim="2d", grn=grn)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(("norm.weight", "norm.bias", "grn_gamma", "grn_beta")):
                p.add_(0.2 * torch.randn_like(p))
    st = {k_: v.detach().clone() for k_, v in m.state_dict().items()}
    assert st["enc_block_0.0.conv1.weight"].dim() == 4 and st["stem.weight"].dim() == 4      # Conv2d parameters
    return m, st


@pytest.mark.parametrize("n_channels,k,norm_type,grn,shape", [
    (8, 3, "group", False, (48, 32)),
    (8, 5, "group", False, (32, 32)),
    (16, 3, "layer", True, (32, 48)),
    (4, 3, "group", True, (16, 16)),
])
def test_mednext_2d_fp32_matches_oracle(n_channels, k, norm_type, grn, shape):
    m, st = _build(n_channels=n_channels, k=k, norm_type=norm_type, grn=grn, ds=True)
    kw = dict(n_channels=n_channels, exp_r=2, kernel_size=k, block_counts=[1] * 9, norm_type=norm_type, grn=grn)
    x = torch.randn(2, 1, *shape, generator=torch.Generator().manual_seed(1))
    ref = MO.forward(st, x, deep_supervision=True, **kw)
    m = m.cuda().eval()
    with torch.no_grad():
        outs = m(x.cuda())
        feat = m.forward_features(x.cuda())
        proj = m.forward_output(feat)
        out5 = m(x.cuda().unsqueeze(2))                       # (B, C, 1, H, W) is accepted too and keeps its rank
    assert isinstance(outs, list) and len(outs) == 5
    for g, r in zip(outs, ref):
        assert g.shape == r.shape and g.dim() == 4
        torch.testing.assert_close(g.cpu(), r, rtol=1e-3, atol=1e-3)
    assert feat.shape == (2, n_channels, *shape)
    torch.testing.assert_close(proj, outs[0], rtol=1e-5, atol=1e-5)
    assert out5[0].shape == (2, 2, 1, *shape)
    torch.testing.assert_close(out5[0].squeeze(2), outs[0], rtol=0, atol=0)
    with pytest.raises(ValueError, match="dim='2d'"):
        m(torch.zeros(1, 1, 2, 32, 32, device="cuda"))
    with pytest.raises(ValueError, match="divisible by 16"):
        m(torch.zeros(1, 1, 24, 32, device="cuda"))


def test_mednext_s_2d_bf16_fused_mixers():
    """The tutorial's topology (MedNeXt-S k3, 2-D, 224 x 224 patches): bf16 storage routes the blocks through the fused
    mixer kernels at depth 1; error budget as for the 3-D network (test_gpu_mednext.py TOL_BF16_PROB)."""
    m, st = _build(n_channels=32, counts=[2] * 9, n_classes=3)
    x = torch.rand(1, 1, 224, 224, generator=torch.Generator().manual_seed(2))
    ref = MO.forward(st, x, n_channels=32, exp_r=2, kernel_size=3, block_counts=[2] * 9)
    m = m.cuda().eval()
    with torch.no_grad():
        got = m(x.cuda()).cpu()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got16 = m(x.cuda()).float().cpu()
    assert got.shape == ref.shape == (1, 3, 224, 224)
    assert (torch.sigmoid(got) - torch.sigmoid(ref)).abs().max() < 1e-3
    assert (torch.sigmoid(got16) - torch.sigmoid(ref)).abs().max() < 4e-2
    assert (torch.sigmoid(got16) - torch.sigmoid(ref)).abs().mean() < 5e-3


@pytest.mark.parametrize("n_channels,k,ds,norm_type,grn", [(8, 3, True, "group", False), (16, 5, False, "group", False),
                                                          (8, 3, False, "layer", True)])
def test_mednext_2d_training_step_matches_oracle_autograd(n_channels, k, ds, norm_type, grn):
    m, st = _build(n_channels=n_channels, k=k, ds=ds, norm_type=norm_type, grn=grn)
    kw = dict(n_channels=n_channels, exp_r=2, kernel_size=k, block_counts=[1] * 9, norm_type=norm_type, grn=grn)
    x = torch.rand(2, 1, 32, 48)
    wmaps = [torch.randn(2, 2, 32 >> i, 48 >> i, generator=torch.Generator().manual_seed(10 + i)) for i in range(5)]

    def objective(out, maps):
        outs = out if isinstance(out, list) else [out]
        return sum((torch.sigmoid(o) * w).mean() for o, w in zip(outs, maps))

    params = {k_: v.clone().requires_grad_(True) for k_, v in st.items() if v.dtype.is_floating_point}
    ref_out = MO.forward(params, x, deep_supervision=ds, **kw)
    objective(ref_out, wmaps).backward()
    m = m.cuda().train()
    out = m(x.cuda())
    objective(out, [w.cuda() for w in wmaps]).backward()
    o0, r0 = (out[0], ref_out[0]) if ds else (out, ref_out)
    assert o0.requires_grad and o0.shape == r0.shape and o0.dim() == 4
    torch.testing.assert_close(o0.detach().cpu(), r0.detach(), rtol=1e-3, atol=1e-3)
    named = dict(m.named_parameters())
    for name, p in params.items():
        if p.grad is None or name == "dummy_tensor":
            continue
        got = named[name].grad
        assert got is not None and got.shape == p.grad.shape, name
        # conv1.bias feeds a per-channel GroupNorm: its true gradient is 0 and both sides hold rounding noise -> floor the scale
        scale = max(p.grad.abs().max().item(), 1e-3 if (name.endswith("conv1.bias") and norm_type == "group") else 1e-5)
        err = (got.cpu() - p.grad).abs().max().item() / scale
        assert err < 2e-2, (name, err, scale)


def test_build_model_2d_multihead_and_module_training_step():
    """build_model with `mednext.dim: 2d` + named task heads; a ConnectomicsModule step on (B, C, H, W) batches with deep
    supervision, fused loss == generic loss path."""
    from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
    from pytorch_connectomics_amd.models import build_model
    from pytorch_connectomics_amd.training.module import ConnectomicsModule
    cfg = NS(model=NS(arch=NS(type="mednext_custom"), in_channels=1, out_channels=2,
                      mednext=NS(base_channels=8, exp_r=2, kernel_size=3, block_counts=[1] * 9, dim="2d"),
                      loss=NS(deep_supervision=False),
                      heads={"aff": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
                             "sdt": {"out_channels": 1, "num_blocks": 0}}, primary_head="aff"))
    torch.manual_seed(0)
    model = build_model(cfg)
    st = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    x = torch.randn(1, 1, 32, 48)
    with torch.no_grad():
        out = model(x.cuda())
    assert out["output"]["aff"].shape == (1, 3, 32, 48) and out["output"]["sdt"].shape == (1, 1, 32, 48)
    trunk = {k[len("model."):]: v for k, v in st.items() if k.startswith("model.")}
    feat = MO.forward_features(trunk, x, n_channels=8, exp_r=2, kernel_size=3, block_counts=[1] * 9)
    hb = {k[len("heads.aff.blocks."):]: v for k, v in st.items() if k.startswith("heads.aff.blocks.")}
    ref_aff = F.conv2d(MO.block_forward(feat, hb, "0", 3), st["heads.aff.projection.weight"], st["heads.aff.projection.bias"])
    ref_sdt = F.conv2d(feat, st["heads.sdt.projection.weight"], st["heads.sdt.projection.bias"])
    torch.testing.assert_close(out["output"]["aff"].cpu(), ref_aff, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out["output"]["sdt"].cpu(), ref_sdt, rtol=1e-3, atol=1e-3)

    c = ConfigNode(schema_defaults())
    c.model.arch.type, c.model.in_channels, c.model.out_channels = "mednext_custom", 1, 2
    c.model.mednext.base_channels, c.model.mednext.exp_r, c.model.mednext.kernel_size = 8, 2, 3
    c.model.mednext.block_counts = [1] * 9
    c.model.mednext.dim = "2d"
    c.model.deep_supervision = True
    c.model.loss.deep_supervision = True
    c.model.loss.losses = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "pos_weight": 2.0},
                           {"function": "DiceLoss", "weight": 0.5, "kwargs": {"sigmoid": True}}]
    torch.manual_seed(3)
    mod = ConnectomicsModule(c).cuda().train()
    xb = torch.rand(2, 1, 32, 48, device="cuda")
    yb = (torch.rand(2, 2, 32, 48, device="cuda") > 0.8).float()
    res = {}
    for fused in (True, False):
        mod.fused_loss = fused
        mod.zero_grad()
        loss = mod.training_step({"image": xb, "label": yb})
        loss.backward()
        res[fused] = (loss.detach().clone(), {n: p.grad.clone() for n, p in mod.model.named_parameters() if p.grad is not None})
    assert torch.isfinite(res[True][0]) and len(res[True][1]) > 50
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    for n, g in res[False][1].items():
        scale = float(g.abs().max().clamp_min(1e-8))
        assert float((res[True][1][n] - g).abs().max()) <= 2e-3 * scale + 1e-7, n


def test_2d_inference_mode_tta_matches_explicit_flips_and_rotations():
    """data.*.do_2d: no sliding window (manager.py:36-44), views over the two image axes (tta_combinations.py:29-34), a
    (B, C, H, W) result.  Against the same views applied by hand around the model."""
    from pytorch_connectomics_amd.inference.manager import InferenceManager
    m, _ = _build(n_channels=8, n_classes=2)
    m = m.cuda().eval()
    cfg = NS(data=NS(train=NS(do_2d=True), val=NS(do_2d=False)),
             model=NS(out_channels=2, output_dtype=None),
             inference=NS(sliding_window=NS(window_size=None, sw_batch_size=1, overlap=0.0, blending="constant", sigma_scale=0.125,
                                            padding_mode="constant", border_mask=None, distributed_sharding=False),
                          test_time_augmentation=NS(enabled=True, flip_axes="all", rotation90_axes=[[0, 1]], rotate90_k=None,
                                                    ensemble_mode="mean", apply_mask=True, patch_first_local=True,
                                                    distributed_sharding=False, distributed_reduce_chunk_mb=128),
                          model=NS(channel_activations=[{"channels": "0:2", "activation": "sigmoid"}], output_head=None,
                                   select_channel=None)))
    mgr = InferenceManager(cfg, m, forward_fn=m)
    assert mgr.sliding_inferer is None
    x = torch.rand(1, 1, 32, 32, device="cuda")
    with torch.no_grad():
        got = mgr.predict_with_tta(x.unsqueeze(2))
        views = []
        for flips in ([], [2], [3], [2, 3]):
            for k in range(4):
                v = torch.flip(x, flips) if flips else x
                v = torch.rot90(v, k, (2, 3))
                p = torch.sigmoid(m(v.contiguous()))
                p = torch.rot90(p, -k, (2, 3))
                views.append(torch.flip(p, flips) if flips else p)
    assert got.shape == (1, 2, 32, 32)
    uniq = []
    for v in views:          # the resolver de-duplicates views that coincide (flip x rot90 has 8 distinct elements)
        if not any(torch.equal(v, u) for u in uniq):
            uniq.append(v)
    assert len(uniq) == 8
    torch.testing.assert_close(got.float(), torch.stack(uniq).mean(0), rtol=1e-5, atol=1e-5)
